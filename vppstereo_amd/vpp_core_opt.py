"""Drop-in for the reference's Cython module ``vpp_core_opt`` (vpp_core/vpp_core_opt.pyx):
same function names, argument order and in-place semantics, executed by the HIP library.

    get_seed()                                        .pyx:23-28   wall-clock seconds (float)
    init_rand(seed)                                   .pyx:33-35
    virtual_projection_scan_rnd(l, r, g, ...)         .pyx:53-54   -> number of hints
    virtual_projection_scan_max_dist(l, r, g, ...)    .pyx:133-134 -> number of hints
    gt_reshape(gt)                                    .pyx:352-371

The libc rand() global state of the reference becomes a (seed, draws consumed) pair kept by
the default context: successive scans continue the same glibc stream exactly like the
reference does between two ``init_rand`` calls.
"""
import ctypes as C

import numpy as np

from . import _lib


def get_seed():
    """Wall-clock seed, CLOCK_REALTIME seconds as a float (.pyx:23-28)."""
    import time
    return time.clock_gettime(time.CLOCK_REALTIME)


def init_rand(_seed=0):
    lib = _lib.load()
    _lib.check(lib.vppx_srand(_lib.default_context().handle, C.c_uint32(int(_seed) & 0xFFFFFFFF)))


def _check(l, r, g, g_occ, width, height, channels):
    for a, dt, nm in ((l, np.uint8, "l"), (r, np.uint8, "r"), (g, np.float32, "g"), (g_occ, np.uint8, "g_occ")):
        if not isinstance(a, np.ndarray) or a.dtype != dt:
            raise ValueError(f"Buffer dtype mismatch for {nm}: expected {np.dtype(dt).name}")
        if not a.flags.c_contiguous:
            raise ValueError(f"{nm} must be C-contiguous")
    if l.ndim != 3 or r.ndim != 3 or g.ndim != 2 or g_occ.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions")
    if l.shape != (height, width, channels) or r.shape != l.shape or g.shape != (height, width) or g_occ.shape != g.shape:
        raise ValueError("shape mismatch between arrays and width/height/channels")


def virtual_projection_scan_rnd(l, r, g, width, height, channels, uniform_color, wsize, direction, c, c_occ, g_occ,
                                discard_occluded, interpolate):
    _check(l, r, g, g_occ, width, height, channels)
    lib = _lib.load()
    rc = lib.vppx_virtual_projection_scan_rnd(_lib.default_context().handle, _lib.np_ptr(l), _lib.np_ptr(r),
                                              _lib.np_ptr(g), int(width), int(height), int(channels),
                                              int(bool(uniform_color)), int(wsize), int(direction), float(c),
                                              float(c_occ), _lib.np_ptr(g_occ), int(bool(discard_occluded)),
                                              int(bool(interpolate)))
    return _lib.check(rc)


def virtual_projection_scan_max_dist(l, r, g, width, height, channels, uniform_color, wsize, wsize_agg_x, wsize_agg_y,
                                     direction, c, c_occ, g_occ, discard_occluded, interpolate):
    _check(l, r, g, g_occ, width, height, channels)
    lib = _lib.load()
    rc = lib.vppx_virtual_projection_scan_max_dist(_lib.default_context().handle, _lib.np_ptr(l), _lib.np_ptr(r),
                                                   _lib.np_ptr(g), int(width), int(height), int(channels),
                                                   int(bool(uniform_color)), int(wsize), int(wsize_agg_x),
                                                   int(wsize_agg_y), int(direction), float(c), float(c_occ),
                                                   _lib.np_ptr(g_occ), int(bool(discard_occluded)),
                                                   int(bool(interpolate)))
    return _lib.check(rc)


def gt_reshape(_gt):
    """Dense hints -> N x 4 (x, y, d, 1) float32 rows in raster order (.pyx:352-371).
    Pure indexing (no arithmetic): done with numpy on the host, like the reference's caller."""
    gt = np.asarray(_gt, np.float32)
    ys, xs = np.nonzero(gt > 0)
    out = np.empty((ys.size, 4), np.float32)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = xs, ys, gt[ys, xs], 1
    return out
