"""Batched, device-resident front-end of the hot path (test.py:158-225 for B frames at once).

torch is plumbing only: tensors own the HBM buffers, ``torch.cuda.current_stream`` orders the
work, ``torch.distributed`` shards frame batches (see dist.py).  Every method hands raw device
pointers to the C-ABI ``_dev`` entry points of libvppx.so.
"""
import ctypes as C

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    """One GPU's VPP+rSGM pipeline.  Inputs are CUDA tensors:
        left, right : uint8  [B,H,W,C]     hints : float32 [B,H,W]     g_occ : uint8 [B,H,W] or None
    """

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("vppstereo_amd.Engine needs an MI355X (no CPU fallback)")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.ctx = _lib.Context(self.device.index)
        self.lib = _lib.load()

    def _bind_stream(self):
        """Run on torch's current stream, so that library kernels are ordered with the torch work around them.
        torch's default stream is the legacy null stream (handle 0): the context then launches on the null stream
        itself (vppx_set_stream_legacy); re-binding an unchanged stream costs nothing."""
        self.ctx.set_stream(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t, dtype, ndim, name, shape=None):
        """Every tensor handed to the C-ABI is checked here: the library trusts B/H/W/C and raw pointers."""
        if t.device != self.device or t.dtype != dtype or t.dim() != ndim or not t.is_contiguous():
            raise ValueError(f"{name}: expected contiguous {dtype} tensor with {ndim} dims on {self.device}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")

    def _chk_pair(self, left, right, hints, g_occ=None):
        torch = self.torch
        self._chk(left, torch.uint8, 4, "left")
        B, H, W, Cc = left.shape
        if Cc not in (1, 3):
            raise ValueError("left: channels must be 1 or 3")
        self._chk(right, torch.uint8, 4, "right", left.shape)
        self._chk(hints, torch.float32, 3, "hints", (B, H, W))
        if g_occ is not None:
            self._chk(g_occ, torch.uint8, 3, "g_occ", (B, H, W))
        return B, H, W, Cc

    def vpp(self, left, right, hints, g_occ=None, seed=1, **kw):
        """In-place-free VPP: returns patterned copies (l_vpp, r_vpp) like vpp() does."""
        B, H, W, Cc = self._chk_pair(left, right, hints, g_occ)
        self._bind_stream()
        l, r = left.clone(), right.clone()
        p = _lib.vpp_params(seed=int(seed) & 0xFFFFFFFF, **kw)
        _lib.check(self.lib.vppx_vpp_dev(self.ctx.handle, C.byref(p), B, H, W, Cc, _ptr(l), _ptr(r), _ptr(hints),
                                         _ptr(g_occ), None, None))
        return l, r

    def rsgm(self, left, left_vpp, right_vpp, out=None, **kw):
        torch = self.torch
        self._chk(left, torch.uint8, 4, "left")
        B, H, W, Cc = left.shape
        self._chk(left_vpp, torch.uint8, 4, "left_vpp", left.shape)
        self._chk(right_vpp, torch.uint8, 4, "right_vpp", left.shape)
        if out is None:
            out = torch.empty((B, H, W), dtype=torch.float32, device=self.device)
        self._chk(out, torch.float32, 3, "out", (B, H, W))
        self._bind_stream()
        p = _lib.rsgm_params(**kw)
        _lib.check(self.lib.vppx_rsgm_dev(self.ctx.handle, C.byref(p), B, H, W, Cc, _ptr(left), _ptr(left_vpp),
                                          _ptr(right_vpp), None, None, _ptr(out)))
        return out

    def rsgm_post(self, disp_l_pad, disp_r_pad, h, w, subpixel=True, out=None):
        """Stage API: the post-processing of compute_rsgm (rsgm.py:275-292) on the padded left / right disparity maps
        [B, Hp, Wp] (Hp, Wp = h, w rounded up to multiples of 16): crop, left/right check, astype(uint8), filterSpeckles(0, 200,
        10), sub-pixel restore, _interpolate_background.  Returns [B, h, w] float32."""
        torch = self.torch
        self._chk(disp_l_pad, torch.float32, 3, "disp_l_pad")
        B, Hp, Wp = disp_l_pad.shape
        self._chk(disp_r_pad, torch.float32, 3, "disp_r_pad", disp_l_pad.shape)
        if (Hp, Wp) != (-(-int(h) // 16) * 16, -(-int(w) // 16) * 16):
            raise ValueError("padded maps must be [B, ceil16(h), ceil16(w)]")
        if out is None:
            out = torch.empty((B, int(h), int(w)), dtype=torch.float32, device=self.device)
        self._chk(out, torch.float32, 3, "out", (B, int(h), int(w)))
        self._bind_stream()
        _lib.check(self.lib.vppx_rsgm_post_dev(self.ctx.handle, B, int(h), int(w), _ptr(disp_l_pad), _ptr(disp_r_pad),
                                               int(bool(subpixel)), _ptr(out)))
        return out

    def vpp_rsgm(self, left, right, hints, g_occ=None, out=None, l_vpp=None, r_vpp=None, seed=1, vpp_kw=None,
                 rsgm_kw=None, occ_out=None, inputs_ready=None):
        """The whole hot path for a batch: one call, no host round trips, no synchronisation.

        g_occ: None, a uint8 [B,H,W] mask tensor, or -- test.py:154 with --maskocc in the same call --
        ``"occlusion_heuristic"`` / a dict of its parameters (rx, ry, l, g, th_conf, th_filter): the mask is then
        computed on the way (vppx_occ_vpp_rsgm_dev) and also written to `occ_out` when given.
        inputs_ready: a ``torch.cuda.Event`` recorded after the work that produced this call's input tensors.  Only
        matters with `set_pipeline(True)`: the front stage then waits for that event instead of for everything queued on
        the current stream, and can overlap the previous call's tail."""
        torch = self.torch
        heuristic = isinstance(g_occ, (str, dict))
        if isinstance(g_occ, str) and g_occ != "occlusion_heuristic":
            raise ValueError("g_occ: expected a mask tensor, None, 'occlusion_heuristic' or a dict of its parameters")
        B, H, W, Cc = self._chk_pair(left, right, hints, None if heuristic else g_occ)
        if out is None:
            out = torch.empty((B, H, W), dtype=torch.float32, device=self.device)
        self._chk(out, torch.float32, 3, "out", (B, H, W))
        for t, n in ((l_vpp, "l_vpp"), (r_vpp, "r_vpp")):
            if t is not None:
                self._chk(t, torch.uint8, 4, n, left.shape)
        if occ_out is not None:
            if not heuristic:
                raise ValueError("occ_out is the mask output of g_occ='occlusion_heuristic'")
            self._chk(occ_out, torch.uint8, 3, "occ_out", (B, H, W))
        self._bind_stream()
        vp = _lib.vpp_params(seed=int(seed) & 0xFFFFFFFF, **(vpp_kw or {}))
        rp = _lib.rsgm_params(**(rsgm_kw or {}))
        if inputs_ready is not None:
            _lib.check(self.lib.vppx_inputs_ready_event(self.ctx.handle, C.c_void_p(inputs_ready.cuda_event)))
        if heuristic:
            op = _lib.occ_params(**(g_occ if isinstance(g_occ, dict) else {}))
            _lib.check(self.lib.vppx_occ_vpp_rsgm_dev(self.ctx.handle, C.byref(op), C.byref(vp), C.byref(rp), B, H, W, Cc,
                                                      _ptr(left), _ptr(right), _ptr(hints), _ptr(occ_out), _ptr(l_vpp),
                                                      _ptr(r_vpp), _ptr(out)))
        else:
            _lib.check(self.lib.vppx_vpp_rsgm_dev(self.ctx.handle, C.byref(vp), C.byref(rp), B, H, W, Cc, _ptr(left),
                                                  _ptr(right), _ptr(hints), _ptr(g_occ), _ptr(l_vpp), _ptr(r_vpp),
                                                  _ptr(out)))
        return out

    def occlusion_heuristic(self, hints, rx=9, ry=7, l=2, g=0.4375, th_conf=1, th_filter=0.1, out=None):
        torch = self.torch
        self._chk(hints, torch.float32, 3, "hints")
        B, H, W = hints.shape
        conf = torch.empty((B, H, W), dtype=torch.uint8, device=self.device) if out is None else out
        self._chk(conf, torch.uint8, 3, "out", (B, H, W))
        self._bind_stream()
        _lib.check(self.lib.vppx_occlusion_heuristic_dev(self.ctx.handle, B, H, W, _ptr(hints), int(rx), int(ry),
                                                         float(l), float(g), float(th_conf), float(th_filter),
                                                         _ptr(conf)))
        return conf

    def to_network_input(self, img_u8, dtype=None, pad_multiple=32):
        """Patterned uint8 [B,H,W,C] -> [B,C,Hq,Wq] float32/bfloat16 in [0,1], replicate-padded to
        multiples of 32: the tensors test.py:179-200 feeds to PSMNet / RAFT-Stereo."""
        torch = self.torch
        self._chk(img_u8, torch.uint8, 4, "img")
        dtype = dtype or torch.float32
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be float32 or bfloat16")
        B, H, W, Cc = img_u8.shape
        ph = (((H // pad_multiple) + 1) * pad_multiple - H) % pad_multiple
        pw = (((W // pad_multiple) + 1) * pad_multiple - W) % pad_multiple
        out = torch.empty((B, Cc, H + ph, W + pw), dtype=dtype, device=self.device)
        self._bind_stream()
        _lib.check(self.lib.vppx_u8_to_nchw_dev(self.ctx.handle, B, H, W, Cc, int(pad_multiple), _ptr(img_u8), _ptr(out),
                                                int(dtype == torch.bfloat16)))
        return out

    # ---- hand-off rows ----
    def psmnet_cost_volume(self, fea_l, fea_r, maxdisp, hints=None, validhints=None):
        """The tensor PSMNet.forward hands to dres0 (models/psmnet/psmnet.py:157-197): concat volume
        [B,2C,maxdisp//4,H4,W4] from the two feature maps [B,C,H4,W4], times the hint modulation when
        hints / validhints ([B,1,H,W] float32) are given."""
        torch = self.torch
        self._chk(fea_l, torch.float32, 4, "fea_l")
        self._chk(fea_r, torch.float32, 4, "fea_r")
        if fea_r.shape != fea_l.shape:
            raise ValueError("feature maps must have the same shape")
        B, Cc, H4, W4 = fea_l.shape
        H = W = 0
        if hints is not None:
            self._chk(hints, torch.float32, 4, "hints")
            self._chk(validhints, torch.float32, 4, "validhints")
            if hints.shape != validhints.shape or hints.shape[:2] != (B, 1):
                raise ValueError("hints / validhints must be [B,1,H,W]")
            H, W = hints.shape[2:]
        cost = torch.empty((B, 2 * Cc, int(maxdisp) // 4, H4, W4), dtype=torch.float32, device=self.device)
        self._bind_stream()
        _lib.check(self.lib.vppx_psmnet_cost_volume_dev(self.ctx.handle, _ptr(fea_l), _ptr(fea_r),
                                                        _ptr(hints) if hints is not None else None,
                                                        _ptr(validhints) if hints is not None else None,
                                                        B, Cc, H4, W4, int(H), int(W), int(maxdisp), _ptr(cost)))
        return cost

    def raft_corr(self, fmap2, fmap3, hints=None, validhints=None):
        """CorrBlock1D.corr (models/raft_stereo/corr.py:151-180): all-pairs correlation along the row
        (a library GEMM through torch) / sqrt(D), then the hint modulation in place on the device."""
        torch = self.torch
        self._chk(fmap2, torch.float32, 4, "fmap2")
        self._chk(fmap3, torch.float32, 4, "fmap3")
        B, Dc, H4, W2 = fmap2.shape
        W3 = fmap3.shape[3]
        corr = torch.einsum('aijk,aijh->ajkh', fmap2, fmap3).reshape(B, H4, W2, 1, W3).contiguous()
        corr = corr / torch.sqrt(torch.tensor(Dc).float())
        if hints is not None:
            self._chk(hints, torch.float32, 4, "hints")
            self._chk(validhints, torch.float32, 4, "validhints")
            self._bind_stream()
            _lib.check(self.lib.vppx_raft_corr_modulate_dev(self.ctx.handle, _ptr(corr), _ptr(hints), _ptr(validhints), B, H4,
                                                            W2, W3, int(hints.shape[2]), int(hints.shape[3])))
        return corr

    def raft_corr_modulate_(self, corr, hints, validhints):
        """In-place hint modulation of an existing correlation volume [B,H4,W2,1,W3] (corr.py:160-178)."""
        torch = self.torch
        self._chk(corr, torch.float32, 5, "corr")
        self._chk(hints, torch.float32, 4, "hints")
        self._chk(validhints, torch.float32, 4, "validhints")
        B, H4, W2, _, W3 = corr.shape
        self._bind_stream()
        _lib.check(self.lib.vppx_raft_corr_modulate_dev(self.ctx.handle, _ptr(corr), _ptr(hints), _ptr(validhints), B, H4, W2,
                                                        W3, int(hints.shape[2]), int(hints.shape[3])))
        return corr

    def kitti_disp_decode(self, png_u16):
        """uint16 PNG samples (device tensor, int16/uint16 storage) -> (disp float32, valid uint8), same shape
        (frame_utils.readDispKITTI :66-69)."""
        torch = self.torch
        if png_u16.dtype not in (torch.int16, torch.uint16) or not png_u16.is_contiguous() or not png_u16.is_cuda:
            raise ValueError("png_u16 must be a contiguous 16-bit device tensor")
        disp = torch.empty(png_u16.shape, dtype=torch.float32, device=self.device)
        valid = torch.empty(png_u16.shape, dtype=torch.uint8, device=self.device)
        self._bind_stream()
        _lib.check(self.lib.vppx_kitti_disp_decode_dev(self.ctx.handle, _ptr(png_u16), png_u16.numel(), _ptr(disp), _ptr(valid)))
        return disp, valid

    def png_decode(self, files, height, width, channels=1, scale=1.0 / 256.0, want="disp"):
        """PNG FILES (a list of bytes objects, e.g. the KITTI disparity / LiDAR maps of frame_utils.readDispKITTI :66-69)
        decoded on the device: one upload of the compressed bytes, then chunk walk, inflate and unfiltering in a
        kernel (one workgroup per file).  want="disp": (disp float32 [n,H,W] = sample * scale, valid uint8);
        want="u8": uint8 [n,H,W,C] samples of 8-bit gray / RGB images.  Raises on a file the decoder rejects."""
        import numpy as np
        torch = self.torch
        n = len(files)
        offs = np.zeros(n + 1, np.int64)
        for i, b in enumerate(files):
            offs[i + 1] = offs[i] + len(b)
        blob = torch.frombuffer(bytearray(b"".join(files)), dtype=torch.uint8).to(self.device)
        status = torch.empty((n,), dtype=torch.int32, device=self.device)
        disp = valid = out = None
        if want == "disp":
            disp = torch.empty((n, height, width), dtype=torch.float32, device=self.device)
            valid = torch.empty((n, height, width), dtype=torch.uint8, device=self.device)
        else:
            out = torch.empty((n, height, width, channels), dtype=torch.uint8, device=self.device)
        self._bind_stream()
        _lib.check(self.lib.vppx_png_decode_dev(self.ctx.handle, n, _ptr(blob), offs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                int(height), int(width), int(channels), float(scale), _ptr(disp), _ptr(valid),
                                                _ptr(out), _ptr(status)))
        st = status.cpu().numpy()
        if st.any():
            bad = int(np.flatnonzero(st)[0])
            raise ValueError(f"png_decode: file {bad} rejected with status {int(st[bad])} (see include/vppx.h)")
        return (disp, valid) if want == "disp" else out

    def pfm_decode(self, raw_u8, height, width, channels=1, little_endian=True):
        """PFM payload bytes (device uint8 tensor, header stripped) -> float32 [H,W] or [H,W,3], flipped
        upside-down like frame_utils.readPFM (:34-64)."""
        torch = self.torch
        self._chk(raw_u8, torch.uint8, 1, "raw")
        if raw_u8.numel() != height * width * channels * 4:
            raise ValueError("payload size does not match height*width*channels*4")
        out = torch.empty((height, width, 3) if channels == 3 else (height, width), dtype=torch.float32, device=self.device)
        self._bind_stream()
        _lib.check(self.lib.vppx_pfm_decode_dev(self.ctx.handle, _ptr(raw_u8), int(height), int(width), int(channels),
                                                int(bool(little_endian)), _ptr(out)))
        return out

    def synchronize(self):
        """Wait for torch's current stream (the stream the library launches on) and check the health of the asynchronous
        hot path: raises `VppxError` when a fused aggregation launch lost its lock step, i.e. when disparities returned
        since the last check are void (include/vppx.h, vppx_status).  Callers that synchronise through torch
        (`torch.cuda.synchronize()`, `.cpu()`, `.item()`) call `status()` afterwards instead."""
        self._bind_stream()
        self.ctx.synchronize()

    def status(self):
        """Non-blocking: raise `VppxError` if any fused aggregation launch finished so far lost its lock step."""
        self.ctx.status()

    def set_pipeline(self, enable=True):
        """Cross-call pipelining for streams of batches through `vpp_rsgm`: the front stage of a call (occlusion
        heuristic, VPP, pad + gray, census) runs on a second stream as soon as the previous call's aggregation is done,
        next to that call's sum / WTA and post kernels.  Safe by construction (include/vppx.h): outputs keep torch's
        stream order and are written on the current stream only (per-call allocated tensors are fine); the front stage
        waits for everything queued on the current stream before the call unless the caller passes `inputs_ready=`,
        the event after which this call's inputs exist -- only then is there anything to overlap.  Off by default."""
        _lib.check(self.lib.vppx_set_pipeline(self.ctx.handle, int(bool(enable))))

    def set_graph_mode(self, enable=True):
        """hipGraph replay of repeated identical vpp_rsgm calls (same tensors, parameters and stream; needs a
        non-default torch stream).  `graph_replays()` counts the calls served by a graph launch."""
        _lib.check(self.lib.vppx_set_graph_mode(self.ctx.handle, int(bool(enable))))

    def graph_replays(self):
        return int(self.lib.vppx_graph_replays(self.ctx.handle))

    # ---- measurement helpers (bench.py) ----
    def time_aggregate(self, iters=10):
        ms = C.c_float()
        _lib.check(self.lib.vppx_time_aggregate(self.ctx.handle, int(iters), C.byref(ms)))
        return float(ms.value)

    def agg_kernel_ms(self, last_n):
        """Average duration of the aggregation kernel over the last `last_n` pipeline launches (hipEvent pairs
        on the launch stream); `last_n <= 0` resets the counter.  Returns (ms, launches averaged)."""
        ms, n = C.c_float(0), C.c_int(0)
        _lib.check(self.lib.vppx_agg_kernel_ms(self.ctx.handle, int(last_n), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def we_kernel_ms(self, last_n):
        """The same for the W/E launch of the fused layout (0 launches in the 8-path layout)."""
        ms, n = C.c_float(0), C.c_int(0)
        _lib.check(self.lib.vppx_we_kernel_ms(self.ctx.handle, int(last_n), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def time_aggregate_frames(self):
        return int(self.lib.vppx_time_aggregate_frames(self.ctx.handle))

    def time_aggregate_part(self, part, iters=10):
        ms = C.c_float()
        _lib.check(self.lib.vppx_time_aggregate_part(self.ctx.handle, int(iters), int(part), C.byref(ms)))
        return float(ms.value)

    def uses_vert(self):
        """Aggregation layout of the last call: 0 = eight line-parallel paths, 1 = band marching (experiment),
        3 = W/E line-parallel + the fused three-path vertical kernel (default from 8 frames per launch on)."""
        return int(self.lib.vppx_uses_vert(self.ctx.handle))

    def last_call_parts(self):
        """Parts the last fused call ran as (a batch larger than one round of the lock-step kernel is split; 1 = unsplit)."""
        return int(self.lib.vppx_last_call_parts(self.ctx.handle))

    def fused_pixels_per_wave(self):
        """16 / 8: which fused vertical kernel the last call used (sgm_vert4_kernel / sgm_vert3_kernel); 0: none."""
        return int(self.lib.vppx_fused_pixels_per_wave(self.ctx.handle))

    def batch_quantum(self, h, w, dmax=192):
        """Frames per full round of the fused layout's lock-step kernel for h x w frames (a scheduling hint: batches that are
        a multiple of it leave no part-filled last round; 0 when the fused layout does not apply)."""
        return int(self.lib.vppx_batch_quantum(self.ctx.handle, int(h), int(w), int(dmax)))

    def enable_stage_timing(self, on=True):
        _lib.check(self.lib.vppx_enable_stage_timing(self.ctx.handle, int(bool(on))))

    def stage_ms(self):
        arr = (C.c_float * 32)()
        n = _lib.check(self.lib.vppx_get_stage_ms(self.ctx.handle, arr, 32))
        return {self.lib.vppx_stage_name(i).decode(): float(arr[i]) for i in range(n)}
