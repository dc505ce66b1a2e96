"""GPU parity of the host-side streaming adapter (vppstereo_amd.pipeline.FrameStream over the C-ABI's vppx_fstream_*): a
stream of numpy frames pushed one at a time equals one `run_frame` call per frame -- test.py:291-311 then :154-225 -- bit for
bit: mask, patterned pair, disparities and the position of the frame's random stream afterwards."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu


def _frames(n, H, W, D, p, seed, gray=False):
    out = []
    for f in range(n):
        fr = synth.make_frame(H, W, D, p, seed=seed + f)
        l, r = (fr["left"][..., 1].copy(), fr["right"][..., 1].copy()) if gray else (fr["left"], fr["right"])
        out.append((l, r, fr["hints"]))
    return out


def _one_by_one(frames, seed, maskocc, vpp_kw, rsgm_kw, g_occ=None):
    """init_rand(seed + f); run_frame(frame f) -- and where the shared libc-like stream stands afterwards."""
    from vppstereo_amd import _lib, pipeline, vpp_standalone
    want = []
    for f, (l, r, h) in enumerate(frames):
        vpp_standalone.init_rand(seed + f)
        d, lc, rc, conf = pipeline.run_frame(l, r, h, maskocc=maskocc, g_occ=None if g_occ is None else g_occ[f], vpp_kw=vpp_kw,
                                             rsgm_kw=rsgm_kw, return_patterns=True)
        s, c = C.c_uint32(), C.c_uint64()
        _lib.check(_lib.load().vppx_rand_state(_lib.default_context().handle, C.byref(s), C.byref(c)))
        want.append((d, lc, rc, conf, int(c.value)))
    return want


@pytest.mark.parametrize("maskocc,depth", [(True, 2), (False, 3), (True, 3), (False, 4)])
def test_stream_of_37_frames_equals_37_run_frame_calls(maskocc, depth):
    """37 frames, batches of 16 (16 + 16 + a flushed 5): every result in input order, equal to the frame's own one-frame
    call with srand(seed + f), including the number of rand() draws it consumed."""
    from vppstereo_amd.pipeline import FrameStream
    H, W, D, seed = 60, 150, 64, 11
    frames = _frames(37, H, W, D, 0.05, 100)
    kw = dict(wsize=5, blending=0.3, c_occ=0.2)
    want = _one_by_one(frames, seed, maskocc, kw, dict(dmax=D, p1=9))
    got, draws = [], []
    # (depth 2: every batch's copy-out right behind its kernels; depth >= 3: held back until the next batch's aggregation is enqueued,
    # released by the pop that needs it when no next batch comes -- the flushed last one here)
    with FrameStream(H, W, 3, batch=16, depth=depth, seed=seed, maskocc=maskocc, vpp_kw=kw, rsgm_kw=dict(dmax=D, p1=9), return_patterns=True) as fs:
        for r in fs.run(iter(frames)):
            got.append(r)
            draws.append(fs.last_draws)
        assert fs.counts()[0] == 37 and fs.counts()[2] == 0 and fs.pending == 0
    assert len(got) == 37
    for f in range(37):
        d, lc, rc, conf = got[f]
        assert np.array_equal(d, want[f][0]), f
        assert np.array_equal(lc, want[f][1]) and np.array_equal(rc, want[f][2]), f
        if maskocc:
            assert np.array_equal(conf, want[f][3]), f
        else:
            assert conf is None
        assert draws[f] == want[f][4] and draws[f] > 0, f


def test_stream_against_the_oracle_with_callers_masks_gray_frames_and_odd_batches():
    """Gray frames, a caller's mask per push, batch 5, depth 3, push / pop by hand with a flush in the middle: the results do
    not depend on where batch boundaries fall, and equal the CPU oracle."""
    from vppstereo_amd.pipeline import FrameStream
    H, W, D, seed = 48, 112, 64, 5
    frames = _frames(13, H, W, D, 0.06, 300, gray=True)
    rng = np.random.default_rng(3)
    masks = [(rng.random((H, W)) < 0.3).astype(np.uint8) for _ in frames]
    got = []
    with FrameStream(H, W, 1, batch=5, depth=3, seed=seed, with_g_occ=True, vpp_kw=dict(c_occ=0.1), rsgm_kw=dict(dmax=D), return_patterns=True) as fs:
        for f, (l, r, h) in enumerate(frames):
            fs.push(l, r, h, masks[f])
            if f == 6:
                fs.flush()              # 5 + 2 | 5 + 1
                assert fs.pending == 7
                for _ in range(3):
                    got.append(fs.pop())
        fs.flush()
        while True:
            r = fs.pop()
            if r is None:
                break
            got.append(r)
        assert fs.pop() is None
    assert len(got) == 13
    for f, (l, r, h) in enumerate(frames):
        oracle.init_rand(seed + f)
        lo, ro = oracle.vpp(l[..., None], r[..., None], h, g_occ=masks[f], c_occ=0.1)
        assert np.array_equal(got[f][1], lo[..., 0]) and np.array_equal(got[f][2], ro[..., 0]), f
        assert np.array_equal(got[f][0], oracle.compute_rsgm(l[..., None], lo, ro, dmax=D)), f


def test_stream_results_do_not_depend_on_batch_size_or_copy_threads():
    from vppstereo_amd.pipeline import FrameStream, run_stream
    H, W, D = 40, 96, 192
    frames = _frames(19, H, W, D, 0.04, 500)
    base = list(run_stream(iter(frames), batch=19, seed=3, maskocc=True, rsgm_kw=dict(dmax=D), copy_threads=1))
    for batch, depth, thr in ((1, 2, 1), (3, 2, 4), (8, 4, -1)):
        out = list(run_stream(iter(frames), batch=batch, depth=depth, seed=3, maskocc=True, rsgm_kw=dict(dmax=D), copy_threads=thr))
        assert len(out) == len(base)
        for a, b in zip(out, base):
            assert np.array_equal(a, b), (batch, depth, thr)
    with FrameStream(H, W, batch=None, rsgm_kw=dict(dmax=D)) as fs:      # the default batch is a whole lock-step round (or 16)
        assert fs.batch >= 1


def test_stream_refuses_what_it_cannot_do_and_reports_a_full_ring():
    from vppstereo_amd import _lib
    from vppstereo_amd.pipeline import FrameStream
    with pytest.raises(ValueError):
        FrameStream(32, 64, maskocc=True, with_g_occ=True)
    with pytest.raises(TypeError):
        FrameStream(32, 64, vpp_kw=dict(blendin=0.2))
    with pytest.raises(Exception, match="dmax % 8"):                      # the hot path's own checks, made when the stream is created
        FrameStream(32, 64, rsgm_kw=dict(dmax=60))
    with pytest.raises(Exception, match="too small"):
        FrameStream(4, 64)
    fr = _frames(1, 32, 64, 64, 0.05, 1)[0]
    with FrameStream(32, 64, batch=2, depth=2, rsgm_kw=dict(dmax=64)) as fs:
        with pytest.raises(ValueError, match="shape"):
            fs.push(fr[0][:-1], fr[1], fr[2])
        with pytest.raises(ValueError, match="with_g_occ"):
            fs.push(*fr, np.zeros((32, 64), np.uint8))
        # the C-ABI itself refuses a push when `depth` whole batches wait to be popped (the Python class pops early instead)
        lib = _lib.load()
        for _ in range(4):
            _lib.check(lib.vppx_fstream_push(fs._h, fr[0].ctypes.data, fr[1].ctypes.data, fr[2].ctypes.data, None))
        rc = lib.vppx_fstream_push(fs._h, fr[0].ctypes.data, fr[1].ctypes.data, fr[2].ctypes.data, None)
        assert rc == -1 and b"pop results first" in lib.vppx_last_error()
        d = np.empty((32, 64), np.float32)
        got = C.c_int(0)
        for _ in range(4):
            _lib.check(lib.vppx_fstream_pop(fs._h, d.ctypes.data, None, None, None, None, C.byref(got)))
            assert got.value == 1
        _lib.check(lib.vppx_fstream_pop(fs._h, d.ctypes.data, None, None, None, None, C.byref(got)))
        assert got.value == 0


@pytest.mark.parametrize("depth,batch,n", [(2, 8, 20), (3, 8, 20), (4, 8, 20), (4, 4, 48), (6, 2, 40)])
def test_stream_reruns_batches_after_a_lost_lock_step(depth, batch, n):
    """VPPX_V3_SPIN_LIMIT=1 makes the fused aggregation give up at the first neighbour record that is not there yet: the stream
    notices at the pop, re-runs what was in flight on the line-parallel layout, and hands out the right disparities."""
    from vppstereo_amd.pipeline import FrameStream
    H, W, D = 40, 96, 192
    frames = _frames(n, H, W, D, 0.04, 700)
    with FrameStream(H, W, batch=batch, seed=9, rsgm_kw=dict(dmax=D)) as ref:
        want = list(ref.run(iter(frames)))
    old = {k: os.environ.get(k) for k in ("VPPX_VERT", "VPPX_V3_SPIN_LIMIT")}
    os.environ["VPPX_VERT"], os.environ["VPPX_V3_SPIN_LIMIT"] = "3", "1"
    try:
        # (ring of 3+: some copy-outs are still held back at the re-run; many small batches in flight: a launch can lose its lock step
        # between the stream's look and the hot path's own -- the submit then takes the report and runs the call again)
        fs = FrameStream(H, W, batch=batch, depth=depth, seed=9, rsgm_kw=dict(dmax=D))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    with fs:
        got = list(fs.run(iter(frames)))
        assert fs.counts()[3] >= 1                      # at least one batch was run again
        assert fs._ctx.lockstep_failures >= 1
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b) and not np.isnan(a).any()


def test_full_size_stream_in_whole_lock_step_rounds_against_the_oracle():
    """26 frames of 540 x 960 x 192 with the default batch (one round of the lock-step kernel: 16 here), occlusion mask on the way: the
    batches take the fused layout with the cut W/E launch, upload / kernels / download of consecutive batches overlap -- and frames 0,
    15, 16 and 25 (first and last of each batch) equal the CPU oracle, every frame its draw count."""
    from vppstereo_amd.pipeline import FrameStream
    H, W, D, seed = 540, 960, 192, 3
    pool = _frames(4, H, W, D, 0.03, 40)
    frames = [pool[(3 * i) % 4] for i in range(26)]
    got, draws = [], []
    with FrameStream(H, W, 3, seed=seed, maskocc=True, rsgm_kw=dict(dmax=D)) as fs:
        assert fs.batch == 16
        for r in fs.run(iter(frames)):
            got.append(r)
            draws.append(fs.last_draws)
        assert fs.counts() == (26, 0, 0, 0)
    assert len(got) == 26
    for f in (0, 15, 16, 25):
        l, r, h = frames[f]
        conf = oracle.occlusion_heuristic(h)[1]
        oracle.init_rand(seed + f)
        lo, ro = oracle.vpp(l, r, h, g_occ=conf)
        assert np.array_equal(got[f], oracle.compute_rsgm(l, lo, ro, dmax=D)), f
    # the same scene at another position of the stream draws from another seed: different colours, so (almost surely) another map
    assert not np.array_equal(got[0], got[4]) and draws[0] > 0 and len(set(draws[i] for i in (0, 4, 8))) == 1


def test_stream_with_the_max_distance_method_and_uniform_colours():
    """The colour method without random draws (vpp_core_opt.pyx:133-341) through the stream: equal to one `run_frame` per frame, draw
    counts 0; uniform colours with the random method: equal as well, one draw per hint and channel."""
    from vppstereo_amd.pipeline import FrameStream
    H, W, D = 40, 104, 64
    frames = _frames(7, H, W, D, 0.05, 900)
    for kw in (dict(method="maxDistance", wsize=3), dict(uniform_color=True, wsize=5)):
        want = _one_by_one(frames, 21, False, dict(kw), dict(dmax=D))
        got, draws = [], []
        with FrameStream(H, W, 3, batch=3, seed=21, vpp_kw=dict(kw), rsgm_kw=dict(dmax=D), return_patterns=True) as fs:
            for r in fs.run(iter(frames)):
                got.append(r)
                draws.append(fs.last_draws)
        for f in range(7):
            assert np.array_equal(got[f][0], want[f][0]) and np.array_equal(got[f][1], want[f][1]) and np.array_equal(got[f][2], want[f][2]), (kw, f)
        if kw.get("method") == "maxDistance":
            assert draws == [0] * 7
        else:
            assert draws == [w[4] for w in want]


@pytest.mark.parametrize("kw", [dict(use_distance_patch=True, wsize=7), dict(use_distance_patch=True, use_bilateral_patch=True, wsize=5, distance_gamma=0.5)])
def test_stream_with_distance_patches_takes_every_frames_own_hint_range(kw):
    """`use_distance_patch`: vpp() takes dmin / dmax from the frame's own hints (vpp_standalone.py:410-411); in a batch every frame
    gets ITS range, computed on the device (VppxVppParams.per_frame_range) -- equal to one `run_frame` per frame, whose host code
    computes the range the reference's way, and for frame 0 to the oracle.  The frames' ranges differ (scaled hints)."""
    from vppstereo_amd.pipeline import FrameStream
    H, W, D = 44, 120, 64
    frames = []
    for f, (l, r, h) in enumerate(_frames(9, H, W, D, 0.05, 1200)):
        frames.append((l, r, (h * np.float32(0.5 + 0.13 * f)).astype(np.float32)))      # another range per frame
    ranges = {(float(h[h > 0].min()), float(h[h > 0].max())) for _, _, h in frames}
    assert len(ranges) == 9
    want = _one_by_one(frames, 5, True, dict(kw), dict(dmax=D))
    got = []
    with FrameStream(H, W, 3, batch=4, seed=5, maskocc=True, vpp_kw=dict(kw), rsgm_kw=dict(dmax=D), return_patterns=True) as fs:
        got = list(fs.run(iter(frames)))
    for f in range(9):
        assert np.array_equal(got[f][1], want[f][1]) and np.array_equal(got[f][2], want[f][2]), f
        assert np.array_equal(got[f][0], want[f][0]) and np.array_equal(got[f][3], want[f][3]), f
    l, r, h = frames[0]
    conf = oracle.occlusion_heuristic(h)[1]
    oracle.init_rand(5)
    lo, ro = oracle.vpp(l, r, h, g_occ=conf, **kw)
    assert np.array_equal(lo, got[0][1]) and np.array_equal(ro, got[0][2])


def test_batched_engine_call_with_per_frame_range_and_a_single_valued_frame():
    """The batched device entry point with per_frame_range: frames of different ranges in one call equal their own one-frame calls;
    a frame whose hints all have ONE value (the reference divides by zero there) gets the full patch size, i.e. equals the same
    frame without use_distance_patch."""
    import torch
    from vppstereo_amd.engine import Engine
    H, W, D = 36, 88, 64
    eng = Engine()
    b = synth.make_batch(4, H, W, D, 0.06, seed=77)
    hints = b["hints"].copy()
    hints[1] *= 0.4
    hints[2] = np.where(hints[2] > 0, np.float32(7.25), np.float32(0))                  # one value only
    dev = eng.device
    L, R, Hn = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (b["left"], b["right"], hints))
    lv, rv = torch.empty_like(L), torch.empty_like(R)
    eng.vpp_rsgm(L, R, Hn, l_vpp=lv, r_vpp=rv, seed=9, vpp_kw=dict(use_distance_patch=1, per_frame_range=1, wsize=7), rsgm_kw=dict(dmax=D))
    eng.synchronize()
    for f in range(4):
        hf = hints[f]
        if f == 2:
            kw = dict(wsize=7)
        else:
            kw = dict(use_distance_patch=1, wsize=7, dmin=float(hf[hf > 0].min()), dmax=float(hf[hf > 0].max()))
        l1, r1 = torch.empty_like(L[f:f + 1]), torch.empty_like(R[f:f + 1])
        eng.vpp_rsgm(L[f:f + 1], R[f:f + 1], Hn[f:f + 1], l_vpp=l1, r_vpp=r1, seed=9 + f, vpp_kw=kw, rsgm_kw=dict(dmax=D))
        eng.synchronize()
        assert torch.equal(l1[0], lv[f]) and torch.equal(r1[0], rv[f]), f
