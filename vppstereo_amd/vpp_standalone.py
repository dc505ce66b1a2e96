"""Drop-in for the reference's ``vpp_standalone.vpp`` (vpp_standalone.py:396-432).

Same signature, defaults, return shapes/dtypes and error behaviour; the scans run on the
MI355X through libvppx.so with the Cython file's arithmetic and glibc random stream (the
parity target named by BASELINE.json; SURVEY.md A.4 lists how the numba twin differs)."""
import ctypes as C

import numpy as np

from . import _lib


def vpp(left, right, gt, wsize=3, wsizeAgg_x=64, wsizeAgg_y=3, left2right=True, blending=0.4,
        use_distance_patch=False, use_bilateral_patch=False, distance_gamma=0.3, bilateral_o_xy=2, bilateral_o_i=1,
        bilateral_th=.001, uniform_color=False, method="rnd", c_occ=0.00, g_occ=None, discard_occ=False,
        interpolate=True):
    lc, rc = np.copy(left), np.copy(right)              # :397 never mutate the arguments
    gt = np.asarray(gt, dtype=np.float32)               # :398 (a copy there; gt is only read here)

    assert method in ["rnd", "maxDistance"]             # :400
    direction = 1 if left2right else 0                  # :401

    if len(lc.shape) < 3:                               # :403-404 gray -> [H,W,1], never squeezed
        lc, rc = np.expand_dims(lc, axis=-1), np.expand_dims(rc, axis=-1)

    dmin, dmax = hint_range(gt)                         # :407, :410-411
    if dmin is None:                                    # :407 no projection without points
        return lc, rc
    if use_distance_patch and not dmax > dmin:
        raise ZeroDivisionError("use_distance_patch needs two distinct hint values "
                                "(vpp_standalone.py:8 divides by dmax-dmin)")

    lc = np.ascontiguousarray(lc, np.uint8)
    rc = np.ascontiguousarray(rc, np.uint8)
    gt = np.ascontiguousarray(gt)
    occ = None
    if g_occ is not None:                               # :424-425 default = no occlusions
        occ = np.ascontiguousarray(np.asarray(g_occ) != 0, np.uint8)
    h, w, ch = lc.shape
    lib = _lib.load()
    ctx = _lib.default_context()
    p = _lib.vpp_params(method=1 if method == "maxDistance" else 0, wsize=int(wsize), wsize_agg_x=int(wsizeAgg_x),
                        wsize_agg_y=int(wsizeAgg_y), direction=direction, uniform_color=int(bool(uniform_color)),
                        discard_occluded=int(bool(discard_occ)), interpolate=int(bool(interpolate)),
                        c=float(blending), c_occ=float(c_occ), use_distance_patch=int(bool(use_distance_patch)),
                        distance_gamma=float(distance_gamma), dmin=dmin, dmax=dmax,
                        use_bilateral_patch=int(bool(use_bilateral_patch)), bilateral_o_xy=float(bilateral_o_xy),
                        bilateral_o_i=float(bilateral_o_i), bilateral_th=float(bilateral_th))
    # continue the libc-like stream of the default context: the SAME (seed, draws consumed) state that
    # vpp_core_opt.init_rand seeds and the single-frame scans advance (global libc state in the reference)
    seed, consumed = C.c_uint32(), C.c_uint64()
    _lib.check(lib.vppx_rand_state(ctx.handle, C.byref(seed), C.byref(consumed)))
    p.seed = seed.value
    p.rand_offset = consumed.value
    nh = (C.c_int64 * 1)()
    _lib.check(lib.vppx_vpp_host(ctx.handle, C.byref(p), 1, h, w, ch, _lib.np_ptr(lc), _lib.np_ptr(rc),
                                 _lib.np_ptr(gt), _lib.np_ptr(occ), None, nh))
    if method == "rnd":  # maxDistance draws nothing from rand()
        draws = (C.c_uint64 * 1)()
        _lib.check(lib.vppx_vpp_last_draws(ctx.handle, 1, draws))
        _lib.check(lib.vppx_rand_advance(ctx.handle, int(draws[0])))
    return lc, rc


def hint_range(gt):
    """vpp_standalone.py:407,410-411 without the temporaries: (None, None) when `np.count_nonzero(gt) == 0`, else
    (min, max) of `gt[gt > 0]` -- ValueError like numpy's min of an empty array when the non-zero hints are all <= 0 (or NaN),
    as there.  Two reductions and one pass over the bit patterns instead of a float count, a boolean index and two more
    reductions (1.6 ms -> 0.25 ms per 540 x 960 map on the build host): positive float32 values order like their bit patterns,
    zeros and negatives are pushed to the top by `bits - 1` as uint32."""
    gt = np.ascontiguousarray(gt, np.float32)
    if gt.size == 0:
        return None, None
    mx, mn = gt.max(), gt.min()
    if mx == 0 and mn == 0:                             # all (+-)0: count_nonzero == 0  (NaN compares false: falls through)
        return None, None
    bits = gt.reshape(-1).view(np.uint32) - np.uint32(1)   # 0 -> 0xFFFFFFFF, negatives (sign bit) stay >= 0x7FFFFFFF
    lo = int(bits.min()) + 1
    if lo > 0x7F800000:                                 # no positive finite-or-inf value at all (+NaN patterns lie above +inf)
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    dmin = float(np.array([lo], np.uint32).view(np.float32)[0])
    if mx != mx:                                        # NaN among the hints: the maximum of the positive ones, the slow way
        pos = gt[gt > 0]
        return float(pos.min()), float(pos.max())
    return dmin, float(mx)


def init_rand(seed=0):
    """Re-seed the stream used by vpp() and by the vpp_core_opt scans (one state, like libc's): same as
    vpp_core_opt.init_rand."""
    from . import vpp_core_opt
    vpp_core_opt.init_rand(seed)
