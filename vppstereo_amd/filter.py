"""Drop-in for the reference's ``filter.occlusion_heuristic`` (filter.py:246-292).

Returns the reference's pair ``(dmap, conf_map)``: element [1] is the uint8 confidence map test.py:154 uses
(1 = occluded/unknown); element [0] is the filtered hint map after ``left_unwarp`` and ``interpolate_disparity(dmap, 3)``.
The reference's ``interpolate_disparity`` indexes ``dmap[y, x +- 1]`` without a bounds test (filter.py:223,229, SURVEY
C-8): column -1 wraps to the last column of the row, column W is the first pixel of the next row, and the read past the
end of the last row -- undefined under numba -- is taken as 0 here."""
import numpy as np

from . import _lib


def occlusion_heuristic(dmap, rx=9, ry=7, l=2, g=0.4375, th_conf=1, th_filter=0.1):
    dmap = np.ascontiguousarray(dmap, np.float32)
    if dmap.ndim != 2:
        raise ValueError("dmap must be HxW")
    h, w = dmap.shape
    conf = np.empty((h, w), np.uint8)
    out = np.empty((h, w), np.float32)
    lib = _lib.load()
    _lib.check(lib.vppx_occlusion_heuristic_full_host(_lib.default_context().handle, 1, h, w, _lib.np_ptr(dmap), int(rx),
                                                      int(ry), float(l), float(g), float(th_conf), float(th_filter),
                                                      _lib.np_ptr(out), _lib.np_ptr(conf)))
    return out, conf
