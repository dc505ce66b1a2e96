"""Drop-in for the reference's ``filter.occlusion_heuristic`` (filter.py:246-292).

test.py:154 only uses element [1] (the uint8 confidence map, 1 = occluded/unknown); element
[0] of the reference is the un-warped hint map after ``interpolate_disparity``, which reads
out of bounds in the reference (SURVEY C-8) and is discarded by its only caller, so ``None``
is returned in its place."""
import numpy as np

from . import _lib


def occlusion_heuristic(dmap, rx=9, ry=7, l=2, g=0.4375, th_conf=1, th_filter=0.1):
    dmap = np.ascontiguousarray(dmap, np.float32)
    if dmap.ndim != 2:
        raise ValueError("dmap must be HxW")
    h, w = dmap.shape
    conf = np.empty((h, w), np.uint8)
    lib = _lib.load()
    _lib.check(lib.vppx_occlusion_heuristic_host(_lib.default_context().handle, 1, h, w, _lib.np_ptr(dmap), int(rx),
                                                 int(ry), float(l), float(g), float(th_conf), float(th_filter),
                                                 _lib.np_ptr(conf)))
    return None, conf
