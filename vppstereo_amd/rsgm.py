"""Drop-in for the reference's ``models/rsgm/rsgm.py``: ``compute_rsgm`` with the same
signature, defaults and exceptions (rsgm.py:250-294); the whole pipeline (pad, census,
8-path aggregation, WTA L/R, sub-pixel, median, gap interpolation, LR check, speckle filter,
background fill) runs on the MI355X."""
import ctypes as C

import numpy as np

from . import _lib


def compute_rsgm(left, left_vpp, right_vpp, hints=None, validhints=None, dmax=192, p1=11, p2min=17, alpha=0.5,
                 gamma=35, uniqueness=0.95, subpixel=True):
    left = np.ascontiguousarray(left, np.uint8)
    left_vpp = np.ascontiguousarray(left_vpp, np.uint8)
    right_vpp = np.ascontiguousarray(right_vpp, np.uint8)
    if left.shape != left_vpp.shape or left.shape != right_vpp.shape:
        raise Exception("left, left_vpp and right_vpp must have the same shape")
    ht, wt = left.shape[:2]
    ch = 1 if left.ndim == 2 else left.shape[2]
    out = np.empty((ht, wt), np.float32)
    hp = vp = None
    if hints is not None and validhints is not None:
        hints = np.ascontiguousarray(hints, np.float32)
        validhints = np.ascontiguousarray(validhints, np.float32)
        hp, vp = _lib.np_ptr(hints), _lib.np_ptr(validhints)
    p = _lib.rsgm_params(dmax=int(dmax), p1=int(p1), p2min=int(p2min), alpha=float(alpha), gamma=int(gamma),
                         uniqueness=float(uniqueness), subpixel=int(bool(subpixel)))
    lib = _lib.load()
    try:
        _lib.check(lib.vppx_rsgm_host(_lib.default_context().handle, C.byref(p), 1, ht, wt, ch, _lib.np_ptr(left),
                                      _lib.np_ptr(left_vpp), _lib.np_ptr(right_vpp), hp, vp, _lib.np_ptr(out)))
    except _lib.VppxError as e:
        raise Exception(str(e)) from e   # the reference raises bare Exception(msg) (rsgm.py:31-40,166)
    return out
