"""test.py:154-225 as ONE call for numpy arrays (a convenience the reference does not have; `filter.occlusion_heuristic`,
`vpp_standalone.vpp` and `rsgm.compute_rsgm` remain the drop-ins for its three call sites).

    disp = run_frame(left, right, hints, maskocc=True)                      # test.py --vpp --maskocc --stereomodel rsgm
    disp, lc, rc, conf = run_frame(..., return_patterns=True)

The pair, the hints and the mask cross PCIe once, the mask never leaves the device
between the occlusion heuristic and the scan, and only the disparity map comes back unless the patterned pair is asked for.
Arithmetic, random stream and results are those of the three separate calls (tests/test_gpu_vpp.py)."""
import ctypes as C

import numpy as np

from . import _lib
from .vpp_standalone import hint_range


def run_frame(left, right, hints, maskocc=False, g_occ=None, occ_kw=None, vpp_kw=None, rsgm_kw=None, return_patterns=False):
    """left, right: uint8 [H,W,3] or [H,W]; hints: float [H,W] (0 = none).  maskocc: compute g_occ with the occlusion heuristic
    (test.py:154) -- else `g_occ` (uint8 [H,W] or None) is used as given.  vpp_kw: keywords of `vpp_standalone.vpp`
    (wsize, blending, c_occ, ...); rsgm_kw: keywords of `rsgm.compute_rsgm` (dmax, p1, ...).  Returns the float32 disparity
    map, or (disparity, left_vpp, right_vpp, conf_map or None) with return_patterns."""
    vk = dict(vpp_kw or {})
    method = vk.pop("method", "rnd")
    assert method in ["rnd", "maxDistance"]
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    gt = np.ascontiguousarray(np.asarray(hints, dtype=np.float32))
    if left.shape != right.shape or left.shape[:2] != gt.shape:
        raise ValueError("left, right and hints must agree in height and width")
    h, w = gt.shape
    ch = 1 if left.ndim == 2 else left.shape[2]
    dmin, dmax = hint_range(gt)
    no_hints = dmin is None
    if no_hints:
        dmin, dmax = 0.0, 0.0     # no hints: the scan leaves the pair as it is (vpp_standalone.py:407)
    p = _lib.vpp_params(method=1 if method == "maxDistance" else 0, wsize=int(vk.pop("wsize", 3)), wsize_agg_x=int(vk.pop("wsizeAgg_x", 64)),
                        wsize_agg_y=int(vk.pop("wsizeAgg_y", 3)), direction=1 if vk.pop("left2right", True) else 0,
                        uniform_color=int(bool(vk.pop("uniform_color", False))), discard_occluded=int(bool(vk.pop("discard_occ", False))),
                        interpolate=int(bool(vk.pop("interpolate", True))), c=float(vk.pop("blending", 0.4)), c_occ=float(vk.pop("c_occ", 0.0)),
                        use_distance_patch=int(bool(vk.pop("use_distance_patch", False))), distance_gamma=float(vk.pop("distance_gamma", 0.3)),
                        dmin=dmin, dmax=dmax, use_bilateral_patch=int(bool(vk.pop("use_bilateral_patch", False))),
                        bilateral_o_xy=float(vk.pop("bilateral_o_xy", 2)), bilateral_o_i=float(vk.pop("bilateral_o_i", 1)),
                        bilateral_th=float(vk.pop("bilateral_th", .001)))
    if vk:
        raise TypeError(f"unknown vpp keyword(s): {sorted(vk)}")
    if no_hints:
        p.use_distance_patch = 0  # vpp() returns the untouched pair before dmin / dmax are looked at (vpp_standalone.py:407)
    if p.use_distance_patch and not dmax > dmin:
        raise ZeroDivisionError("use_distance_patch needs two distinct hint values (vpp_standalone.py:8 divides by dmax-dmin)")
    rk = dict(rsgm_kw or {})
    if "subpixel" in rk:
        rk["subpixel"] = int(bool(rk["subpixel"]))
    rp = _lib.rsgm_params(**rk)
    op = _lib.occ_params(**(occ_kw or {})) if maskocc else None
    occ = None
    if not maskocc and g_occ is not None:
        occ = np.ascontiguousarray(np.asarray(g_occ) != 0, np.uint8)
    lib, ctx = _lib.load(), _lib.default_context()
    seed, consumed = C.c_uint32(), C.c_uint64()
    _lib.check(lib.vppx_rand_state(ctx.handle, C.byref(seed), C.byref(consumed)))   # the stream vpp() and the scans share
    p.seed, p.rand_offset = seed.value, consumed.value
    disp = np.empty((h, w), np.float32)
    lc = rc = conf = None
    if return_patterns:
        lc, rc = np.empty((h, w, ch), np.uint8), np.empty((h, w, ch), np.uint8)
        conf = np.empty((h, w), np.uint8) if maskocc else None
    draws = (C.c_uint64 * 1)()
    try:
        _lib.check(lib.vppx_occ_vpp_rsgm_host(ctx.handle, C.byref(op) if op is not None else None, C.byref(p), C.byref(rp), 1, h, w, ch,
                                              _lib.np_ptr(left), _lib.np_ptr(right), _lib.np_ptr(gt), _lib.np_ptr(occ), _lib.np_ptr(conf),
                                              _lib.np_ptr(lc), _lib.np_ptr(rc), _lib.np_ptr(disp), draws))
    except _lib.VppxError as e:
        raise Exception(str(e)) from e   # the reference raises bare Exception(msg) (rsgm.py:31-40,166)
    if method == "rnd":
        _lib.check(lib.vppx_rand_advance(ctx.handle, int(draws[0])))
    return (disp, lc, rc, conf) if return_patterns else disp
