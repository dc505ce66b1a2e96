"""Drop-in for the reference's ``pyrSGM`` extension (import at models/rsgm/rsgm.py:6): the
seven natives under their own names, caller-allocated outputs written in place, executed by
HIP kernels.  Call sites: rsgm.py:25,26,44,61,141,142,145,170,173."""
import numpy as np

from . import _lib


def _h():
    return _lib.load(), _lib.default_context().handle


def census5x5_SSE(img, out, w, h):
    lib, ctx = _h()
    assert img.dtype == np.uint8 and out.dtype == np.uint32 and img.flags.c_contiguous and out.flags.c_contiguous
    _lib.check(lib.vppx_census5x5(ctx, _lib.np_ptr(img), _lib.np_ptr(out), int(w), int(h)))


def costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, w, h, dmax, n_threads=1):
    lib, ctx = _h()
    assert cl.dtype == np.uint32 and cr.dtype == np.uint32 and dsi.dtype == np.uint16
    _lib.check(lib.vppx_cost_census5x5_xyd(ctx, _lib.np_ptr(cl), _lib.np_ptr(cr), _lib.np_ptr(dsi), int(w), int(h),
                                           int(dmax), int(n_threads)))


def aggregate_SSE(img, dsi, dsi_agg, w, h, dmax, p1, p2min, alpha, gamma):
    """call site rsgm.py:61.  The reference's glue passes the padded H x W x 3 colour `left` here (rsgm.py:258,270)
    although the native reads one byte per pixel; a 3-channel image is converted to gray on the device (the P2
    image of the fused compute_rsgm), [H,W] and [H,W,1] are taken as they are, anything else is rejected."""
    lib, ctx = _h()
    assert img.dtype == np.uint8 and dsi.dtype == np.uint16 and dsi_agg.dtype == np.uint16
    img = np.ascontiguousarray(img)
    if img.shape[:2] != (int(h), int(w)) or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] not in (1, 3)):
        raise Exception(f"aggregate_SSE: image must be {h}x{w} with 1 or 3 channels, got shape {img.shape}")
    ch = 3 if (img.ndim == 3 and img.shape[2] == 3) else 1
    _lib.check(lib.vppx_aggregate_img(ctx, _lib.np_ptr(img), ch, _lib.np_ptr(dsi), _lib.np_ptr(dsi_agg), int(w), int(h),
                                      int(dmax), int(p1), int(p2min), float(alpha), int(gamma)))


def matchWTA_SSE(dsi, disp, w, h, dmax, uniqueness):
    lib, ctx = _h()
    assert dsi.dtype == np.uint16 and disp.dtype == np.float32
    _lib.check(lib.vppx_match_wta(ctx, _lib.np_ptr(dsi), _lib.np_ptr(disp), int(w), int(h), int(dmax),
                                  float(uniqueness)))


def matchWTARight_SSE(dsi, disp, w, h, dmax, uniqueness):
    lib, ctx = _h()
    assert dsi.dtype == np.uint16 and disp.dtype == np.float32
    _lib.check(lib.vppx_match_wta_right(ctx, _lib.np_ptr(dsi), _lib.np_ptr(disp), int(w), int(h), int(dmax),
                                        float(uniqueness)))


def subPixelRefine(dsi, disp, w, h, dmax, method):
    lib, ctx = _h()
    assert dsi.dtype == np.uint16 and disp.dtype == np.float32
    _lib.check(lib.vppx_subpixel_refine(ctx, _lib.np_ptr(dsi), _lib.np_ptr(disp), int(w), int(h), int(dmax),
                                        int(method)))


def median3x3_SSE(src, dst, w, h):
    lib, ctx = _h()
    assert src.dtype == np.float32 and dst.dtype == np.float32
    _lib.check(lib.vppx_median3x3(ctx, _lib.np_ptr(src), _lib.np_ptr(dst), int(w), int(h)))
