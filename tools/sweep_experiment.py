"""Experiment driver for the raster-order four-path sweep (rsgm_kernels.hip: sweep4_kernel): parity of one
pass against the oracle's path subsets on a small frame, then timing at 544x960x192 for a batch."""
import ctypes as C
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (tools/ experiment only)
import synth  # noqa: E402
from vppstereo_amd import _lib  # noqa: E402

lib, ctx = _lib.load(), _lib.default_context()
lib.vppx_debug_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int,
                                                                                                C.c_void_p, C.POINTER(C.c_float)]


def run(gray, cl, cr, mirror, copies=1):
    h, w = gray.shape
    out = np.zeros((h, w, 192), np.uint8)
    ms = C.c_float()
    _lib.check(lib.vppx_debug_sweep(ctx.handle, _lib.np_ptr(gray), _lib.np_ptr(cl), _lib.np_ptr(cr), w, h, 192, 11, 17, 0.5, 35, mirror,
                                    copies, _lib.np_ptr(out), C.byref(ms)))
    return out, ms.value


def scene(H, W, seed):
    fr = synth.make_frame(H, W, 192, 0.05, seed=seed)
    gl, gr = oracle.rgb2gray(fr["left"]), oracle.rgb2gray(fr["right"])
    cl, cr = np.zeros((H, W), np.uint32), np.zeros((H, W), np.uint32)
    oracle.census5x5_SSE(gl, cl, W, H)
    oracle.census5x5_SSE(gr, cr, W, H)
    return gl, cl, cr


for (H, W) in ((16, 32), (32, 208), (48, 240)):
    gl, cl, cr = scene(H, W, 3)
    dsi = np.zeros((H, W, 192), np.uint16)
    oracle.costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, W, H, 192, 1)
    for mirror, mask in ((0, 0x0F), (1, 0xF0)):
        S = np.zeros_like(dsi)
        oracle.aggregate_SSE(gl, dsi, S, W, H, 192, 11, 17, 0.5, 35, path_mask=mask)
        got, _ = run(gl, cl, cr, mirror)
        bad = int((got.astype(np.uint16) != S).sum())
        print(f"{H}x{W} mirror={mirror}: mismatching cells {bad} / {S.size}", flush=True)
        if bad:
            yy, xx, dd = np.nonzero(got.astype(np.uint16) != S)
            print("  first:", yy[:5], xx[:5], dd[:5], got[yy[0], xx[0], dd[0]], S[yy[0], xx[0], dd[0]])

if len(sys.argv) > 1:
    B = int(sys.argv[1])
    gl, cl, cr = scene(544, 960, 5)
    for mirror in (0, 1):
        _, ms = run(gl, cl, cr, mirror, copies=B)
        print(f"544x960x192 B={B} mirror={mirror}: {ms:.3f} ms per pass", flush=True)
