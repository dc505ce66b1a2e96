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


def _vpp_params(vk, method, dmin, dmax):
    """Keywords of `vpp_standalone.vpp` (vpp_standalone.py:396-399) -> VppxVppParams; `vk` is consumed, leftovers raise."""
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
    return p


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
    p = _vpp_params(vk, method, dmin, dmax)
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


class FrameStream:
    """The hot path for a caller that holds ONE numpy frame at a time (test.py:291-311: a DataLoader with batch size 1, then
    :154-225 per frame) at the rate of the batched kernels: frames go in one at a time, results come out one at a time in
    input order, and in between the library batches them (libvppx `vppx_fstream_*`: one multi-threaded copy into a page-locked
    ring of `depth` batches, upload of batch k+1 under batch k's kernels, copy-out under batch k+1's sum / WTA kernel; depth 3 is
    what keeps the device busy, depth 2 saves a third of the ring's page-locked and device memory at ~10 % of the rate).

        with FrameStream(540, 960, maskocc=True, rsgm_kw=dict(dmax=192), seed=7) as fs:
            for disp in fs.run((l, r, hints) for l, r, hints in loader):     # input order
                ...
        # or: fs.push(l, r, hints); ...; fs.flush(); r = fs.pop()

    Frame f (counted from the stream's creation) draws its colours from srand(seed + f): its results equal
    `init_rand(seed + f); run_frame(...)` bit for bit whatever `batch`, wherever a flush falls.  (The reference's single libc
    stream running through all frames is a serial dependency between frames; per-frame streams are what lets frames be
    batched and sharded)  `use_distance_patch`: dmin / dmax are each frame's own, computed on the device
    (a frame whose hints all have ONE value, where the reference divides by zero, gets the full patch size).  batch=None: one round of the lock-step kernel for this shape (vppx_batch_quantum), else 16.
    Results: the float32 disparity map, or with return_patterns (disparity, left_vpp, right_vpp, conf_map or None); each a fresh
    array.  `draws` of the last popped frame (rand() calls it consumed) is in `last_draws`."""

    def __init__(self, height, width, channels=3, batch=None, depth=3, seed=1, maskocc=False, with_g_occ=False, occ_kw=None,
                 vpp_kw=None, rsgm_kw=None, return_patterns=False, copy_threads=-1, device=-1):
        vk = dict(vpp_kw or {})
        method = vk.pop("method", "rnd")
        assert method in ["rnd", "maxDistance"]
        if maskocc and with_g_occ:
            raise ValueError("either maskocc (the mask is computed on the way) or with_g_occ (every push brings one)")
        p = _vpp_params(vk, method, 0.0, 0.0)
        p.seed = int(seed) & 0xFFFFFFFF
        rk = dict(rsgm_kw or {})
        if "subpixel" in rk:
            rk["subpixel"] = int(bool(rk["subpixel"]))
        rp = _lib.rsgm_params(**rk)
        op = _lib.occ_params(**(occ_kw or {})) if maskocc else None
        self.h, self.w, self.ch = int(height), int(width), int(channels)
        self._shape = (self.h, self.w) if self.ch == 1 else (self.h, self.w, self.ch)
        self._lib = _lib.load()
        self._ctx = _lib.Context(device)
        if batch is None:
            batch = int(self._lib.vppx_batch_quantum(self._ctx.handle, self.h, self.w, int(rp.dmax))) or 16
        self.batch, self.depth = int(batch), int(depth)
        self._patterns, self._mask, self._gocc = bool(return_patterns), bool(maskocc and return_patterns), bool(with_g_occ)
        flags = (_lib.FS_PATTERNS if self._patterns else 0) | (_lib.FS_MASK if self._mask else 0) | (_lib.FS_GOCC if self._gocc else 0)
        self._h = C.c_void_p()
        try:
            _lib.check(self._lib.vppx_fstream_create(self._ctx.handle, C.byref(op) if op is not None else None, C.byref(p), C.byref(rp),
                                                     self.batch, self.depth, self.h, self.w, self.ch, flags, int(copy_threads), C.byref(self._h)))
        except _lib.VppxError as e:
            self._ctx.close()
            raise Exception(str(e)) from e
        self._pushed = self._popped = self._fill = 0
        self._inflight = []       # frames not yet popped of every submitted batch, oldest first (at most `depth` batches)
        self._early = []          # results popped to make room, oldest first
        self.last_draws = 0

    # ---- context manager / lifetime ----
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vppx_fstream_destroy(self._h)
            self._h = C.c_void_p()
            self._ctx.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one frame in ----
    def _arr(self, a, dtype, shape, name):
        a = np.asarray(a)
        if a.dtype != dtype or not a.flags.c_contiguous:
            a = np.ascontiguousarray(a, dtype)
        if a.shape != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {a.shape}")
        return a

    def push(self, left, right, hints, g_occ=None):
        left = self._arr(left, np.uint8, self._shape, "left")
        right = self._arr(right, np.uint8, self._shape, "right")
        hints = self._arr(hints, np.float32, (self.h, self.w), "hints")
        occ = None
        if self._gocc:
            if g_occ is None:
                raise ValueError("the stream was created with with_g_occ=True: every push brings a mask")
            occ = np.ascontiguousarray(np.asarray(g_occ) != 0, np.uint8)
            if occ.shape != (self.h, self.w):
                raise ValueError("g_occ: expected shape (H, W)")
        elif g_occ is not None:
            raise ValueError("g_occ given but the stream was created without with_g_occ=True")
        # the ring holds `depth` submitted batches: before the frame that would start one more, take the oldest batch's results
        if self._fill == 0:
            while len(self._inflight) >= self.depth:
                self._early.append(self._pop_native())
        try:
            _lib.check(self._lib.vppx_fstream_push(self._h, left.ctypes.data, right.ctypes.data, hints.ctypes.data,
                                                   occ.ctypes.data if occ is not None else None))
        except _lib.VppxError as e:
            raise Exception(str(e)) from e
        self._pushed += 1
        self._fill += 1
        if self._fill == self.batch:
            self._inflight.append(self._fill)
            self._fill = 0

    def flush(self):
        """Submit the frames pushed so far as a smaller batch (end of the sequence, or a latency bound)."""
        try:
            _lib.check(self._lib.vppx_fstream_flush(self._h))
        except _lib.VppxError as e:
            raise Exception(str(e)) from e
        if self._fill:
            self._inflight.append(self._fill)
            self._fill = 0

    # ---- one frame out ----
    def _pop_native(self):
        disp = np.empty((self.h, self.w), np.float32)
        lc = rc = conf = None
        if self._patterns:
            lc, rc = np.empty(self._shape, np.uint8), np.empty(self._shape, np.uint8)
        if self._mask:
            conf = np.empty((self.h, self.w), np.uint8)
        draws, got = C.c_uint64(0), C.c_int(0)
        try:
            _lib.check(self._lib.vppx_fstream_pop(self._h, disp.ctypes.data, lc.ctypes.data if lc is not None else None,
                                                  rc.ctypes.data if rc is not None else None, conf.ctypes.data if conf is not None else None,
                                                  C.byref(draws), C.byref(got)))
        except _lib.VppxError as e:
            raise Exception(str(e)) from e
        if not got.value:
            return None
        self._popped += 1
        self._inflight[0] -= 1
        if self._inflight[0] == 0:
            self._inflight.pop(0)
        return ((disp, lc, rc, conf) if self._patterns else disp), int(draws.value)

    def pop(self):
        """The next result in input order (waits for its batch), or None when nothing submitted is outstanding."""
        r = self._early.pop(0) if self._early else self._pop_native()
        if r is None:
            return None
        self.last_draws = r[1]
        return r[0]

    @property
    def pending(self):
        """Frames pushed whose results have not been popped."""
        return self._pushed - self._popped + len(self._early)

    def counts(self):
        """(frames pushed, frames of the batch being filled, frames submitted and not popped, batches re-run after a lost lock step)."""
        v = [C.c_int64(0) for _ in range(4)]
        _lib.check(self._lib.vppx_fstream_counts(self._h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def run(self, frames):
        """Generator: results for an iterable of (left, right, hints[, g_occ]) in input order; keeps about `depth` batches in flight."""
        for fr in frames:
            self.push(*fr)
            # whole batches that must be complete by now: everything but the last depth - 1 submitted ones
            while self._early:
                yield self.pop()
            while len(self._inflight) >= self.depth:
                yield self.pop()
        self.flush()
        while True:
            r = self.pop()
            if r is None:
                return
            yield r


def run_stream(frames, **kw):
    """`FrameStream(...).run(frames)` with the shape taken from the first frame; keywords as for `FrameStream`."""
    it = iter(frames)
    try:
        first = next(it)
    except StopIteration:
        return
    l = np.asarray(first[0])
    ch = 1 if l.ndim == 2 else l.shape[2]

    def chain():
        yield first
        yield from it
    with FrameStream(l.shape[0], l.shape[1], ch, with_g_occ=kw.pop("with_g_occ", len(first) > 3), **kw) as fs:
        yield from fs.run(chain())
