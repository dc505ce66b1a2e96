"""ctypes binding of libvppx.so (C-ABI in include/vppx.h).

There is no CPU fallback anywhere in this package: if the HIP library is missing or no
MI355X is visible, every compute call raises.  PyTorch is used only as plumbing (device
memory, streams, torch.distributed) by the optional tensor front-end.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VPPX_LIB") or os.path.join(_HERE, "libvppx.so")  # VPPX_LIB: experiment builds

VPPX_OK = 0
ERR_NAMES = {
    -1: "VPPX_E_INVALID_ARG", -2: "VPPX_E_DMAX_MOD8", -3: "VPPX_E_DMAX_GT256", -4: "VPPX_E_UNIQUENESS",
    -5: "VPPX_E_WIDTH_MOD16", -6: "VPPX_E_METHOD", -7: "VPPX_E_NO_DEVICE", -8: "VPPX_E_HIP", -9: "VPPX_E_OOM",
    -10: "VPPX_E_UNSUPPORTED",
}

# every symbol include/vppx.h declares (tests check that the library exports all of them)
EXPORTS = [
    "vppx_version", "vppx_last_error", "vppx_vpp_params_default", "vppx_rsgm_params_default", "vppx_occ_params_default", "vppx_create",
    "vppx_destroy", "vppx_set_stream", "vppx_set_stream_legacy", "vppx_set_pipeline", "vppx_synchronize", "vppx_status", "vppx_lockstep_failures", "vppx_workspace_bytes", "vppx_device_name", "vppx_srand",
    "vppx_rand_stream", "vppx_rand_state", "vppx_rand_advance", "vppx_virtual_projection_scan_rnd", "vppx_virtual_projection_scan_max_dist", "vppx_vpp_host",
    "vppx_vpp_dev", "vppx_vpp_last_draws", "vppx_census5x5", "vppx_cost_census5x5_xyd", "vppx_aggregate", "vppx_aggregate_img", "vppx_match_wta",
    "vppx_match_wta_right", "vppx_subpixel_refine", "vppx_median3x3", "vppx_rsgm_host", "vppx_rsgm_dev",
    "vppx_vpp_rsgm_dev", "vppx_rsgm_post_dev", "vppx_occ_vpp_rsgm_dev", "vppx_occ_vpp_rsgm_host", "vppx_inputs_ready_event", "vppx_u8_to_nchw_dev", "vppx_psmnet_cost_volume_dev", "vppx_raft_corr_modulate_dev", "vppx_kitti_disp_decode_dev", "vppx_png_decode_dev", "vppx_pfm_decode_dev", "vppx_occlusion_heuristic_host", "vppx_occlusion_heuristic_dev", "vppx_occlusion_heuristic_full_host", "vppx_occlusion_heuristic_full_dev", "vppx_set_graph_mode", "vppx_graph_replays", "vppx_time_aggregate", "vppx_agg_kernel_ms", "vppx_we_kernel_ms", "vppx_time_aggregate_frames", "vppx_time_aggregate_part", "vppx_uses_vert", "vppx_last_call_parts", "vppx_fused_pixels_per_wave", "vppx_batch_quantum",
    "vppx_enable_stage_timing", "vppx_get_stage_ms", "vppx_stage_name",
    "vppx_fstream_create", "vppx_fstream_destroy", "vppx_fstream_push", "vppx_fstream_flush", "vppx_fstream_pop", "vppx_fstream_counts",
]
FS_PATTERNS, FS_MASK, FS_GOCC = 1, 2, 4


class VppxVppParams(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("wsize", C.c_int32), ("wsize_agg_x", C.c_int32), ("wsize_agg_y", C.c_int32),
        ("direction", C.c_int32), ("uniform_color", C.c_int32), ("discard_occluded", C.c_int32),
        ("interpolate", C.c_int32), ("c", C.c_float), ("c_occ", C.c_float), ("use_distance_patch", C.c_int32),
        ("use_bilateral_patch", C.c_int32), ("distance_gamma", C.c_double), ("dmin", C.c_float), ("dmax", C.c_float),
        ("seed", C.c_uint32), ("per_frame_range", C.c_uint32), ("rand_offset", C.c_uint64),
        ("bilateral_o_xy", C.c_double), ("bilateral_o_i", C.c_double), ("bilateral_th", C.c_double),
    ]


class VppxRsgmParams(C.Structure):
    _fields_ = [
        ("dmax", C.c_int32), ("p1", C.c_int32), ("p2min", C.c_int32), ("alpha", C.c_float), ("gamma", C.c_int32),
        ("uniqueness", C.c_float), ("subpixel", C.c_int32), ("reserved0", C.c_int32),
    ]


class VppxOccParams(C.Structure):
    _fields_ = [("rx", C.c_int32), ("ry", C.c_int32), ("l", C.c_double), ("g", C.c_double), ("th_conf", C.c_double),
                ("th_filter", C.c_double)]


class VppxError(Exception):
    """Raised for every non-zero return of the C-ABI; .code is the VPPX_E_* value.  The
    message is the reference's own exception text where one exists (rsgm.py:19-40)."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_lib = None
_lock = threading.Lock()


def load():
    """Load libvppx.so; raises (never falls back) when it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C vppstereo_amd/csrc`). "
                "vppstereo_amd has no CPU fallback.")
        # torch bundles its own HIP runtime (same soname as /opt/rocm's): whichever is loaded first
        # serves the whole process, and torch only works with its own -> let torch go first.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.vppx_version.restype = C.c_int
        lib.vppx_last_error.restype = C.c_char_p
        lib.vppx_device_name.restype = C.c_char_p
        lib.vppx_device_name.argtypes = [vp]
        lib.vppx_stage_name.restype = C.c_char_p
        lib.vppx_stage_name.argtypes = [C.c_int]
        lib.vppx_workspace_bytes.restype = C.c_size_t
        lib.vppx_workspace_bytes.argtypes = [vp]
        lib.vppx_create.argtypes = [C.POINTER(vp), C.c_int]
        lib.vppx_destroy.argtypes = [vp]
        lib.vppx_destroy.restype = None
        lib.vppx_set_stream.argtypes = [vp, vp]
        lib.vppx_set_stream_legacy.argtypes = [vp]
        lib.vppx_set_pipeline.argtypes = [vp, C.c_int]
        lib.vppx_synchronize.argtypes = [vp]
        lib.vppx_status.argtypes = [vp]
        lib.vppx_lockstep_failures.argtypes = [vp]
        lib.vppx_lockstep_failures.restype = C.c_long
        lib.vppx_srand.argtypes = [vp, C.c_uint32]
        lib.vppx_rand_state.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        lib.vppx_rand_advance.argtypes = [vp, C.c_uint64]
        lib.vppx_rand_stream.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_int64, vp]
        i, f, d = C.c_int, C.c_float, C.c_double
        lib.vppx_virtual_projection_scan_rnd.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, f, f, vp, i, i]
        lib.vppx_virtual_projection_scan_max_dist.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, f, vp, i, i]
        pv, pr = C.POINTER(VppxVppParams), C.POINTER(VppxRsgmParams)
        lib.vppx_vpp_params_default.argtypes = [pv]
        lib.vppx_vpp_params_default.restype = None
        lib.vppx_rsgm_params_default.argtypes = [pr]
        lib.vppx_rsgm_params_default.restype = None
        lib.vppx_vpp_host.argtypes = [vp, pv, i, i, i, i, vp, vp, vp, vp, vp, vp]
        lib.vppx_vpp_dev.argtypes = [vp, pv, i, i, i, i, vp, vp, vp, vp, vp, vp]
        lib.vppx_vpp_last_draws.argtypes = [vp, i, C.POINTER(C.c_uint64)]
        lib.vppx_census5x5.argtypes = [vp, vp, vp, i, i]
        lib.vppx_cost_census5x5_xyd.argtypes = [vp, vp, vp, vp, i, i, i, i]
        lib.vppx_aggregate.argtypes = [vp, vp, vp, vp, i, i, i, i, i, f, i]
        lib.vppx_aggregate_img.argtypes = [vp, vp, i, vp, vp, i, i, i, i, i, f, i]
        lib.vppx_match_wta.argtypes = [vp, vp, vp, i, i, i, f]
        lib.vppx_match_wta_right.argtypes = [vp, vp, vp, i, i, i, f]
        lib.vppx_subpixel_refine.argtypes = [vp, vp, vp, i, i, i, i]
        lib.vppx_median3x3.argtypes = [vp, vp, vp, i, i]
        lib.vppx_rsgm_host.argtypes = [vp, pr, i, i, i, i, vp, vp, vp, vp, vp, vp]
        lib.vppx_rsgm_dev.argtypes = [vp, pr, i, i, i, i, vp, vp, vp, vp, vp, vp]
        lib.vppx_rsgm_post_dev.argtypes = [vp, i, i, i, vp, vp, i, vp]
        lib.vppx_vpp_rsgm_dev.argtypes = [vp, pv, pr, i, i, i, i, vp, vp, vp, vp, vp, vp, vp]
        po = C.POINTER(VppxOccParams)
        lib.vppx_occ_vpp_rsgm_host.argtypes = [vp, po, pv, pr, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_uint64)]
        lib.vppx_occ_params_default.argtypes = [po]
        lib.vppx_occ_params_default.restype = None
        lib.vppx_occ_vpp_rsgm_dev.argtypes = [vp, po, pv, pr, i, i, i, i, vp, vp, vp, vp, vp, vp, vp]
        lib.vppx_inputs_ready_event.argtypes = [vp, vp]
        lib.vppx_u8_to_nchw_dev.argtypes = [vp, i, i, i, i, i, vp, vp, i]
        lib.vppx_psmnet_cost_volume_dev.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp]
        lib.vppx_raft_corr_modulate_dev.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i]
        lib.vppx_kitti_disp_decode_dev.argtypes = [vp, vp, C.c_int64, vp, vp]
        lib.vppx_pfm_decode_dev.argtypes = [vp, vp, i, i, i, i, vp]
        lib.vppx_png_decode_dev.argtypes = [vp, i, vp, C.POINTER(C.c_int64), i, i, i, f, vp, vp, vp, vp]
        lib.vppx_occlusion_heuristic_host.argtypes = [vp, i, i, i, vp, i, i, d, d, d, d, vp]
        lib.vppx_occlusion_heuristic_dev.argtypes = [vp, i, i, i, vp, i, i, d, d, d, d, vp]
        lib.vppx_occlusion_heuristic_full_host.argtypes = [vp, i, i, i, vp, i, i, d, d, d, d, vp, vp]
        lib.vppx_occlusion_heuristic_full_dev.argtypes = [vp, i, i, i, vp, i, i, d, d, d, d, vp, vp]
        lib.vppx_set_graph_mode.argtypes = [vp, i]
        lib.vppx_graph_replays.argtypes = [vp]
        lib.vppx_graph_replays.restype = C.c_long
        lib.vppx_time_aggregate.argtypes = [vp, i, C.POINTER(C.c_float)]
        lib.vppx_agg_kernel_ms.argtypes = [vp, i, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        lib.vppx_we_kernel_ms.argtypes = [vp, i, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        lib.vppx_time_aggregate_frames.argtypes = [vp]
        lib.vppx_time_aggregate_part.argtypes = [vp, i, i, C.POINTER(C.c_float)]
        lib.vppx_uses_vert.argtypes = [vp]
        lib.vppx_last_call_parts.argtypes = [vp]
        lib.vppx_fused_pixels_per_wave.argtypes = [vp]
        lib.vppx_batch_quantum.argtypes = [vp, i, i, i]
        lib.vppx_enable_stage_timing.argtypes = [vp, i]
        lib.vppx_get_stage_ms.argtypes = [vp, C.POINTER(C.c_float), i]
        lib.vppx_fstream_create.argtypes = [vp, po, pv, pr, i, i, i, i, i, i, i, C.POINTER(vp)]
        lib.vppx_fstream_destroy.argtypes = [vp]
        lib.vppx_fstream_destroy.restype = None
        lib.vppx_fstream_push.argtypes = [vp, vp, vp, vp, vp]
        lib.vppx_fstream_flush.argtypes = [vp]
        lib.vppx_fstream_pop.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        p64 = C.POINTER(C.c_int64)
        lib.vppx_fstream_counts.argtypes = [vp, p64, p64, p64, p64]
        _lib = lib
        return lib


def check(rc):
    """Turn a negative return code into the reference's `raise Exception(msg)`."""
    if rc is not None and rc < 0:
        msg = load().vppx_last_error().decode("utf-8", "replace")
        raise VppxError(rc, msg or ERR_NAMES.get(rc, f"vppx error {rc}"))
    return rc


def np_ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class Context:
    """One vppx_ctx (one GPU, one stream, one workspace arena)."""

    def __init__(self, device=-1):
        self._lib = load()
        self._h = C.c_void_p()
        check(self._lib.vppx_create(C.byref(self._h), int(device)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self._lib.vppx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        """Bind a hipStream_t handle; 0 / None = the legacy default stream (torch's default stream)."""
        if not stream_ptr:
            check(self._lib.vppx_set_stream_legacy(self._h))
        else:
            check(self._lib.vppx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_own_stream(self):
        """Back to the context's private non-blocking stream (what a fresh context uses)."""
        check(self._lib.vppx_set_stream(self._h, None))

    def synchronize(self):
        """Wait for the context's stream; raises VppxError when a fused aggregation launch lost its lock step."""
        check(self._lib.vppx_synchronize(self._h))

    def status(self):
        """Non-blocking health check (vppx_status): raises VppxError for a lost lock step seen so far."""
        check(self._lib.vppx_status(self._h))

    @property
    def lockstep_failures(self):
        return int(self._lib.vppx_lockstep_failures(self._h))

    @property
    def device_name(self):
        return self._lib.vppx_device_name(self._h).decode()

    @property
    def workspace_bytes(self):
        return int(self._lib.vppx_workspace_bytes(self._h))


_default_ctx = None


def default_context():
    """Process-wide context used by the drop-in module-level functions (the reference's
    natives are free functions with global libc rand() state)."""
    global _default_ctx
    if _default_ctx is None:
        dev = -1
        lr = os.environ.get("LOCAL_RANK")
        if lr is not None:
            try:
                import torch
                if torch.cuda.device_count() > int(lr):
                    dev = int(lr)
            except Exception:
                dev = -1
        _default_ctx = Context(dev)
    return _default_ctx


def _fill(p, kw):
    # setattr on a ctypes.Structure silently creates a Python attribute for an unknown name: a misspelt or
    # reference-style keyword (blending=, wsizeAgg_x=) must not be ignored
    names = {f[0] for f in p._fields_ if not f[0].startswith("reserved")}
    for k, v in kw.items():
        if k not in names:
            raise TypeError(f"{type(p).__name__}: unknown parameter {k!r} (fields: {sorted(names)})")
        setattr(p, k, v)
    return p


def vpp_params(**kw):
    p = VppxVppParams()
    load().vppx_vpp_params_default(C.byref(p))
    return _fill(p, kw)


def occ_params(**kw):
    p = VppxOccParams()
    load().vppx_occ_params_default(C.byref(p))
    return _fill(p, kw)


def rsgm_params(**kw):
    p = VppxRsgmParams()
    load().vppx_rsgm_params_default(C.byref(p))
    return _fill(p, kw)


def c_contig(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a
