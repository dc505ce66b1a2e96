"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; the product path
(``vppstereo_amd``) never does.

The functions mirror the reference's Python-visible names so parity tests read like
the reference's call sites:
  * ``virtual_projection_scan_rnd`` / ``virtual_projection_scan_max_dist`` /
    ``init_rand``                       -- vpp_core/vpp_core_opt.pyx:33,53,133
  * ``vpp``                             -- vpp_standalone.py:396-432
  * ``census5x5_SSE`` ... ``median3x3_SSE`` -- pyrSGM names, call sites rsgm.py:25-173
  * ``compute_rsgm``                    -- models/rsgm/rsgm.py:250-294
  * ``occlusion_heuristic``             -- filter.py:246-292
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
# ORACLE_LIB: use another build of the same sources, e.g. the ASan/UBSan one (`make -C oracle asan`; CPU only)
_LIB_OVERRIDE = os.environ.get("ORACLE_LIB")


def build(force=False):
    """Compile oracle/liboracle.so with gcc (Makefile in this directory)."""
    srcs = [os.path.join(_HERE, f) for f in ("vpp_oracle.c", "rsgm_oracle.c", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if _LIB_OVERRIDE:
            _lib = C.CDLL(_LIB_OVERRIDE)
        else:
            build()
            _lib = C.CDLL(_LIB_PATH)
        _lib.vppo_rand.restype = C.c_int
        _lib.rsgmo_compute_rsgm.restype = C.c_int
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _u8(a):
    return _p(a, C.c_uint8)


def _f32(a):
    return _p(a, C.c_float)


# --------------------------------------------------------------------------- rand
def init_rand(seed=0):
    """vpp_core_opt.pyx:33-35 -> srand((int)seed)"""
    lib().vppo_srand(C.c_uint(int(seed) & 0xFFFFFFFF))


def rand():
    return lib().vppo_rand()


def rand_stream(seed, n):
    out = np.empty(n, np.int32)
    lib().vppo_rand_stream(C.c_uint(int(seed) & 0xFFFFFFFF), C.c_int64(n), _p(out, C.c_int32))
    return out


# --------------------------------------------------------------------------- VPP scans
def _chk_scan(l, r, g, g_occ):
    assert l.dtype == np.uint8 and r.dtype == np.uint8 and l.flags.c_contiguous and r.flags.c_contiguous
    assert g.dtype == np.float32 and g.flags.c_contiguous
    assert g_occ.dtype == np.uint8 and g_occ.flags.c_contiguous
    assert l.ndim == 3 and l.shape == r.shape and g.shape == l.shape[:2] == g_occ.shape


def virtual_projection_scan_rnd(l, r, g, width, height, channels, uniform_color, wsize, direction, c, c_occ,
                                g_occ, discard_occluded, interpolate, filled_g=None, use_distance_patch=False,
                                dmin=0.0, dmax=0.0, distance_gamma=0.3):
    """In place on l, r; returns the number of hints (vpp_core_opt.pyx:53-131)."""
    _chk_scan(l, r, g, g_occ)
    fg = None if filled_g is None else _f32(np.ascontiguousarray(filled_g, np.float32))
    return lib().vppo_scan_rnd_ex(_u8(l), _u8(r), _f32(g), int(width), int(height), int(channels),
                                  int(bool(uniform_color)), int(wsize), int(direction), C.c_float(c),
                                  C.c_float(c_occ), _u8(g_occ), int(bool(discard_occluded)),
                                  int(bool(interpolate)), fg, int(bool(use_distance_patch)), C.c_float(dmin),
                                  C.c_float(dmax), C.c_double(distance_gamma))


def virtual_projection_scan_max_dist(l, r, g, width, height, channels, uniform_color, wsize, wsize_agg_x,
                                     wsize_agg_y, direction, c, c_occ, g_occ, discard_occluded, interpolate,
                                     filled_g=None, use_distance_patch=False, dmin=0.0, dmax=0.0,
                                     distance_gamma=0.3):
    """In place on l, r; returns the number of hints (vpp_core_opt.pyx:133-341)."""
    _chk_scan(l, r, g, g_occ)
    fg = None if filled_g is None else _f32(np.ascontiguousarray(filled_g, np.float32))
    return lib().vppo_scan_max_dist_ex(_u8(l), _u8(r), _f32(g), int(width), int(height), int(channels),
                                       int(bool(uniform_color)), int(wsize), int(wsize_agg_x), int(wsize_agg_y),
                                       int(direction), C.c_float(c), C.c_float(c_occ), _u8(g_occ),
                                       int(bool(discard_occluded)), int(bool(interpolate)), fg,
                                       int(bool(use_distance_patch)), C.c_float(dmin), C.c_float(dmax),
                                       C.c_double(distance_gamma))


def patch_radius(d_ref, d_min, d_max, patch_size, gamma=0.3):
    """vpp_standalone.py:7-11 (numba typing)"""
    f = lib().vppo_patch_radius
    return f(C.c_float(d_ref), C.c_float(d_min), C.c_float(d_max), int(patch_size), C.c_double(gamma))


def rgb2gray(img, bgr=False):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty(img.shape[:2], np.uint8)
    lib().rsgmo_rgb2gray(_u8(img), _u8(out), C.c_int64(out.size), int(bool(bgr)))
    return out


def bilateral_filling(dmap, img, n, o_xy=2, o_i=1, th=.001):
    """vpp_standalone.py:372-394"""
    dmap = np.ascontiguousarray(dmap, np.float32)
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(dmap)
    h, w = dmap.shape
    lib().vppo_bilateral_filling(_f32(dmap), _u8(img), h, w, int(n), C.c_double(o_xy), C.c_double(o_i),
                                 C.c_double(th), _f32(out))
    return out


def vpp(left, right, gt, wsize=3, wsizeAgg_x=64, wsizeAgg_y=3, left2right=True, blending=0.4,
        use_distance_patch=False, use_bilateral_patch=False, distance_gamma=0.3, bilateral_o_xy=2, bilateral_o_i=1,
        bilateral_th=.001, uniform_color=False, method="rnd", c_occ=0.00, g_occ=None, discard_occ=False,
        interpolate=True):
    """vpp_standalone.py:396-432 with the Cython scan arithmetic (the oracle of record).

    The libc-style random stream continues from the last ``init_rand``."""
    lc, rc = np.copy(left), np.copy(right)
    gt = gt.astype(np.float32)
    assert method in ["rnd", "maxDistance"]
    direction = 1 if left2right else 0
    if len(lc.shape) < 3:
        lc, rc = np.expand_dims(lc, axis=-1), np.expand_dims(rc, axis=-1)
    if np.count_nonzero(gt) == 0:
        return lc, rc
    dmin = gt[gt > 0].min()
    dmax = gt[gt > 0].max()
    if use_distance_patch and not dmax > dmin:
        # _get_patch_size_based_on_distance (vpp_standalone.py:6-11) is @njit with numba's default error model: the float
        # division by d_max - d_min == 0 raises there, for the first hint it meets
        raise ZeroDivisionError("division by zero")
    if len(lc.shape) == 3 and lc.shape[2] == 3:
        gray_context = rgb2gray(lc, bgr=True)
    else:
        gray_context = np.squeeze(lc, axis=-1)
    filled_gt = None
    if use_bilateral_patch:
        filled_gt = bilateral_filling(gt, gray_context, (wsize - 1) // 2, bilateral_o_xy, bilateral_o_i, bilateral_th)
    if g_occ is None:
        g_occ = np.zeros_like(gt)
    g_occ = np.ascontiguousarray(g_occ != 0, np.uint8)
    lc, rc, gt = np.ascontiguousarray(lc), np.ascontiguousarray(rc), np.ascontiguousarray(gt)
    h, w, ch = lc.shape
    if method == "maxDistance":
        virtual_projection_scan_max_dist(lc, rc, gt, w, h, ch, uniform_color, wsize, wsizeAgg_x, wsizeAgg_y,
                                         direction, blending, c_occ, g_occ, discard_occ, interpolate, filled_gt,
                                         use_distance_patch, dmin, dmax, distance_gamma)
    else:
        virtual_projection_scan_rnd(lc, rc, gt, w, h, ch, uniform_color, wsize, direction, blending, c_occ, g_occ,
                                    discard_occ, interpolate, filled_gt, use_distance_patch, dmin, dmax,
                                    distance_gamma)
    return lc, rc


# --------------------------------------------------------------------------- pyrSGM natives
def census5x5_SSE(img, out, w, h):
    lib().rsgmo_census5x5(_u8(img), _p(out, C.c_uint32), int(w), int(h))


def costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, w, h, dmax, nthreads=1):
    lib().rsgmo_cost_census5x5_xyd(_p(cl, C.c_uint32), _p(cr, C.c_uint32), _p(dsi, C.c_uint16), int(w), int(h),
                                   int(dmax))


def aggregate_SSE(img, dsi, dsi_agg, w, h, dmax, p1, p2min, alpha, gamma, path_mask=0xFF, simd=False):
    """simd=True: the AVX2 twin (rsgmo_aggregate_paths_simd), bit-equal; the scalar function is the checker."""
    assert img.dtype == np.uint8 and img.shape == (h, w)
    fn = lib().rsgmo_aggregate_paths_simd if simd else lib().rsgmo_aggregate_paths
    fn(_u8(img), _p(dsi, C.c_uint16), _p(dsi_agg, C.c_uint16), int(w), int(h), int(dmax),
       int(p1), int(p2min), C.c_float(alpha), int(gamma), int(path_mask))


def set_simd(on):
    """compute_rsgm aggregates and takes its winners with the AVX2 twins from here on (bench.py's `cpu_baseline_simd` leg only; off by default)."""
    lib().rsgmo_set_simd(int(bool(on)))


def get_simd():
    return bool(lib().rsgmo_get_simd())


def _guided_dsi(dsi, hints, validhints):
    """rsgm.py:116-127: returns the re-weighted copy (uint16)."""
    out = np.ascontiguousarray(dsi, np.uint16).copy()
    h, w, dmax = out.shape
    lib().rsgmo_guided_dsi(_p(out, C.c_uint16), _f32(np.ascontiguousarray(hints, np.float32)),
                           _f32(np.ascontiguousarray(validhints, np.float32)), int(w), int(h), int(dmax))
    return out


def matchWTA_SSE(dsi, disp, w, h, dmax, uniqueness, simd=False):
    """simd=True: the AVX2 twin (rsgmo_match_wta_simd), bit-equal; the scalar function is the checker."""
    if simd:
        lib().rsgmo_match_wta_simd(_p(dsi, C.c_uint16), _f32(disp), int(w), int(h), int(dmax), C.c_float(uniqueness), 0)
    else:
        lib().rsgmo_match_wta(_p(dsi, C.c_uint16), _f32(disp), int(w), int(h), int(dmax), C.c_float(uniqueness))


def matchWTARight_SSE(dsi, disp, w, h, dmax, uniqueness, simd=False):
    if simd:
        lib().rsgmo_match_wta_simd(_p(dsi, C.c_uint16), _f32(disp), int(w), int(h), int(dmax), C.c_float(uniqueness), 1)
    else:
        lib().rsgmo_match_wta_right(_p(dsi, C.c_uint16), _f32(disp), int(w), int(h), int(dmax), C.c_float(uniqueness))


def subPixelRefine(dsi, disp, w, h, dmax, method):
    lib().rsgmo_subpixel_refine(_p(dsi, C.c_uint16), _f32(disp), int(w), int(h), int(dmax), int(method))


def median3x3_SSE(src, dst, w, h, simd=False):
    (lib().rsgmo_median3x3_simd if simd else lib().rsgmo_median3x3)(_f32(src), _f32(dst), int(w), int(h))


def p2_lut(p2min, alpha, gamma):
    out = np.empty(256, np.int32)
    lib().rsgmo_p2_lut(int(p2min), C.c_float(alpha), int(gamma), _p(out, C.c_int32))
    return out


def pad_reflect(img, top, bottom, left, right):
    img = np.ascontiguousarray(img, np.uint8)
    a = img if img.ndim == 3 else img[..., None]
    h, w, c = a.shape
    out = np.empty((h + top + bottom, w + left + right, c), np.uint8)
    lib().rsgmo_pad_reflect_u8(_u8(a), h, w, c, top, bottom, left, right, _u8(out))
    return out if img.ndim == 3 else out[..., 0]


# --------------------------------------------------------------------------- rsgm.py glue
def _linear_interpolate(dmap, n=3, th=1):
    assert dmap.dtype == np.float32 and dmap.flags.c_contiguous
    h, w = dmap.shape
    lib().rsgmo_linear_interpolate(_f32(dmap), h, w, int(n), C.c_double(th))


def _left_right_check(dl, dr, th=1):
    dl = np.ascontiguousarray(dl, np.float32)
    dr = np.ascontiguousarray(dr, np.float32)
    h, w = dl.shape
    mask = np.empty((h, w), np.uint8)
    lib().rsgmo_left_right_check(_f32(dl), _f32(dr), _u8(mask), h, w, C.c_double(th))
    return mask


def _interpolate_background(dmap):
    assert dmap.dtype == np.float32 and dmap.flags.c_contiguous
    h, w = dmap.shape
    lib().rsgmo_interpolate_background(_f32(dmap), h, w)


def filterSpeckles(img, new_val, max_size, max_diff):
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    h, w = img.shape
    lib().rsgmo_filter_speckles_u8(_u8(img), h, w, int(new_val), int(max_size), int(max_diff))


def compute_rsgm(left, left_vpp, right_vpp, hints=None, validhints=None, dmax=192, p1=11, p2min=17, alpha=0.5,
                 gamma=35, uniqueness=0.95, subpixel=True, return_padded=False):
    """models/rsgm/rsgm.py:250-294"""
    left = np.ascontiguousarray(left, np.uint8)
    left_vpp = np.ascontiguousarray(left_vpp, np.uint8)
    right_vpp = np.ascontiguousarray(right_vpp, np.uint8)
    ht, wt = left.shape[:2]
    c = 1 if left.ndim == 2 else left.shape[2]
    out = np.empty((ht, wt), np.float32)
    hp = vp = None
    if hints is not None and validhints is not None:
        hints = np.ascontiguousarray(hints, np.float32)
        validhints = np.ascontiguousarray(validhints, np.float32)
        hp, vp = _f32(hints), _f32(validhints)
    pad_h = (((ht // 16) + 1) * 16 - ht) % 16
    pad_w = (((wt // 16) + 1) * 16 - wt) % 16
    dl = np.empty((ht + pad_h, wt + pad_w), np.float32)
    dr = np.empty_like(dl)
    rc = lib().rsgmo_compute_rsgm(_u8(left), _u8(left_vpp), _u8(right_vpp), ht, wt, c, hp, vp, int(dmax), int(p1),
                                  int(p2min), C.c_float(alpha), int(gamma), C.c_float(uniqueness),
                                  int(bool(subpixel)), _f32(out), _f32(dl), _f32(dr))
    if rc == -2:
        raise Exception(f"Invalid dmax ({dmax}): dmax % 8 != 0")
    if rc == -3:
        raise Exception(f"Invalid dmax ({dmax}): dmax > 256")
    if rc == -4:
        raise Exception(f"Invalid uniqueness ({uniqueness}): uniqueness in ]0,1]")
    if return_padded:
        return out, dl, dr
    return out


# --------------------------------------------------------------------------- filter.py
def occlusion_heuristic(dmap, rx=9, ry=7, l=2, g=0.4375, th_conf=1, th_filter=0.1):
    """filter.py:246-292 -> (dmap, conf_map)"""
    dmap = np.ascontiguousarray(dmap, np.float32)
    h, w = dmap.shape
    dout = np.empty_like(dmap)
    conf = np.empty((h, w), np.uint8)
    lib().flto_occlusion_heuristic(_f32(dmap), h, w, int(rx), int(ry), C.c_double(l), C.c_double(g),
                                   C.c_double(th_conf), C.c_double(th_filter), _f32(dout), _u8(conf))
    return dout, conf


def guided_metrics(disp, gt, valid):
    """losses.py:13-24"""
    disp = np.ascontiguousarray(disp, np.float32)
    gt = np.ascontiguousarray(gt, np.float32)
    valid = np.ascontiguousarray(valid, np.float32)
    out = np.empty(6, np.float64)
    lib().rsgmo_guided_metrics(_f32(disp), _f32(gt), _f32(valid), C.c_int64(disp.size), _p(out, C.c_double))
    return dict(zip(['bad 1.0', 'bad 2.0', 'bad 3.0', 'bad 4.0', 'avgerr', 'rms'], out.tolist()))
