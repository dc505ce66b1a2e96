"""GPU: the launch shape bench.py times, tested as it is timed -- 32 frames per call at full size, the occlusion mask
from occlusion_heuristic, cross-call pipelining on, consecutive steps without a synchronisation in between, 32
DISTINCT scenes -- plus the failure path of the fused aggregation kernel: a lost lock step must be reported by the
synchronisation of the SAME call (vppx_synchronize / vppx_status), never surface as wrong disparities with rc 0
(the reference's aggregate_SSE, rsgm.py:61, is synchronous and cannot do that), and the context must carry on,
correctly, on the line-parallel kernel."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu


def _engine(**env):
    """A fresh Engine created under the given environment (the context reads its knobs at creation)."""
    from vppstereo_amd.engine import Engine
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            os.environ[k] = str(v)
        return Engine()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _dev(eng, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


@pytest.mark.parametrize("H,W,p", [(540, 960, 0.03), (375, 1242, 0.05)])
def test_benchmarked_launch_shape_b32_mask_pipelined_distinct_scenes(H, W, p):
    """B = 32, D = 192, g_occ from occlusion_heuristic, set_pipeline(True), three consecutive steps with nothing but the
    calls themselves between them: every frame of every step equals the unpipelined 8-path layout bit for bit, the
    masks and patterned pairs too, and the first four frames equal the CPU oracle."""
    import torch
    B, D, STEPS = 32, 192, 3
    eng = _engine()
    eng.set_pipeline(True)
    ref = _engine(VPPX_VERT=0)
    b = synth.make_batch(B, H, W, D, p, seed=4242 + H)
    left, right, hints = (_dev(eng, b[k]) for k in ("left", "right", "hints"))
    occ_buf = torch.empty((B, H, W), dtype=torch.uint8, device=eng.device)
    outs = [torch.empty((B, H, W), dtype=torch.float32, device=eng.device) for _ in range(STEPS)]
    lvs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=eng.device) for _ in range(STEPS)]
    rvs = [torch.empty_like(lvs[0]) for _ in range(STEPS)]
    torch.cuda.synchronize()
    ev_ready = torch.cuda.Event()   # the inputs are resident (bench.py says the same to the library)
    ev_ready.record()
    torch.cuda.synchronize()
    for s in range(STEPS):      # the bench loop (bench.py local_step): no synchronisation between the steps
        eng.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", occ_out=occ_buf, out=outs[s], l_vpp=lvs[s], r_vpp=rvs[s],
                     seed=11 + 100 * s, rsgm_kw=dict(dmax=D, subpixel=1), inputs_ready=ev_ready)
    eng.synchronize()           # raises if a fused launch lost its lock step
    assert eng.uses_vert() == 3
    occ_ref = ref.occlusion_heuristic(hints)
    for s in range(STEPS):
        lv = torch.empty_like(lvs[0])
        rv = torch.empty_like(lvs[0])
        want = ref.vpp_rsgm(left, right, hints, g_occ=occ_ref, l_vpp=lv, r_vpp=rv, seed=11 + 100 * s, rsgm_kw=dict(dmax=D, subpixel=1))
        ref.synchronize()
        assert ref.uses_vert() == 0
        assert torch.equal(occ_buf, occ_ref)
        assert torch.equal(lvs[s], lv) and torch.equal(rvs[s], rv), s
        bad = (outs[s] != want).flatten(1).any(1).nonzero().flatten().tolist()
        assert not bad, (s, bad)
    got, glv, grv = outs[0].cpu().numpy(), lvs[0].cpu().numpy(), rvs[0].cpu().numpy()
    occ_np = occ_buf.cpu().numpy()
    for f in (0, 1, 17, 31):
        conf = oracle.occlusion_heuristic(b["hints"][f])[1]
        assert np.array_equal(conf, occ_np[f]), f
        oracle.init_rand(11 + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f], g_occ=conf)
        assert np.array_equal(lo, glv[f]) and np.array_equal(ro, grv[f]), f
        assert np.array_equal(oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D, subpixel=True), got[f]), f


def _small_batch(eng, B=8, H=135, W=240, D=192, seed=9):
    b = synth.make_batch(B, H, W, D, 0.04, seed=seed)
    return b, [_dev(eng, b[k]) for k in ("left", "right", "hints")]


def test_forced_lockstep_timeout_is_reported_by_the_same_calls_sync():
    """VPPX_V3_SPIN_LIMIT=1 makes every wave of the fused kernel give up at the first neighbour record that is not there
    yet.  The synchronisation of that very call must report it; the next call runs the line-parallel layout and is
    right; nothing is reported twice."""
    import torch
    from vppstereo_amd._lib import VppxError
    D = 192
    eng = _engine(VPPX_VERT=3, VPPX_V3_SPIN_LIMIT=1)
    ref = _engine(VPPX_VERT=0)
    b, args = _small_batch(eng)
    want = ref.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D))
    ref.synchronize()
    lost = eng.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D))
    assert eng.uses_vert() == 3
    with pytest.raises(VppxError, match="lost its lock step") as ei:
        eng.synchronize()
    assert ei.value.code == -8          # VPPX_E_HIP
    # ... and nothing queued behind the call could have mistaken its output for disparities: it is NaN throughout
    assert bool(torch.isnan(lost).all())
    assert eng.ctx.lockstep_failures == 1
    eng.synchronize()                   # reported once
    out = eng.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D))
    eng.synchronize()
    assert eng.uses_vert() == 0         # the context stays on the line-parallel kernel
    assert torch.equal(out, want)
    oracle.init_rand(3)
    lo, ro = oracle.vpp(b["left"][0], b["right"][0], b["hints"][0])
    assert np.array_equal(oracle.compute_rsgm(b["left"][0], lo, ro, dmax=D), out[0].cpu().numpy())


def test_lost_lock_step_rests_on_the_line_kernel_then_tries_the_fused_layout_again():
    """Contention is usually temporary: after a loss the context aggregates with the line-parallel kernel for 64 launches
    (doubling with every loss), then tries the fused layout again.  With the forced give-up the retry loses again -- and says
    so again, once."""
    import torch
    from vppstereo_amd._lib import VppxError
    D = 192
    eng = _engine(VPPX_VERT=3, VPPX_V3_SPIN_LIMIT=1)
    ref = _engine(VPPX_VERT=0)
    b, args = _small_batch(eng, B=8, H=40, W=96, seed=15)
    want = ref.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D)).clone()
    ref.synchronize()
    eng.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D))
    with pytest.raises(VppxError, match="lost its lock step.*next 64 aggregation launches"):
        eng.synchronize()
    for i in range(64):
        out = eng.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D))
        assert eng.uses_vert() == 0, i
    eng.synchronize()
    assert torch.equal(out, want) and eng.ctx.lockstep_failures == 1
    lost = eng.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D))     # the rest is over: fused again
    assert eng.uses_vert() == 3
    with pytest.raises(VppxError, match="next 128 aggregation launches"):
        eng.synchronize()
    assert eng.ctx.lockstep_failures == 2 and bool(torch.isnan(lost).all())
    out = eng.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=D))
    eng.synchronize()
    assert eng.uses_vert() == 0 and torch.equal(out, want)


def test_forced_lockstep_timeout_is_seen_by_status_and_by_the_next_call():
    """A caller that synchronises through torch asks vppx_status; one that does not ask at all gets the error from its
    next call into the hot path (before anything is queued)."""
    import torch
    from vppstereo_amd._lib import VppxError
    D = 192
    ref = _engine(VPPX_VERT=0)
    eng = _engine(VPPX_VERT=3, VPPX_V3_SPIN_LIMIT=1)
    b, args = _small_batch(eng, seed=10)
    want = ref.vpp_rsgm(*args, seed=4, rsgm_kw=dict(dmax=D))
    ref.synchronize()
    eng.status()                        # nothing has run: healthy
    eng.vpp_rsgm(*args, seed=4, rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()
    with pytest.raises(VppxError, match="lost its lock step"):
        eng.status()
    eng.status()
    eng2 = _engine(VPPX_VERT=3, VPPX_V3_SPIN_LIMIT=1)
    eng2.vpp_rsgm(*args, seed=4, rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()
    with pytest.raises(VppxError, match="lost its lock step"):
        eng2.occlusion_heuristic(args[2])
    for e in (eng, eng2):
        out = e.vpp_rsgm(*args, seed=4, rsgm_kw=dict(dmax=D))
        e.synchronize()
        assert e.uses_vert() == 0 and torch.equal(out, want)


def test_forced_lockstep_timeout_host_entry_point_recovers_by_itself():
    """vppx_rsgm_host is synchronous: it sees the mark of its own aggregation and repeats it on the line-parallel
    kernel, so a host-pointer caller (the compute_rsgm drop-in) never receives void disparities."""
    from vppstereo_amd import _lib
    B, H, W, D = 4, 64, 160, 192
    old = {k: os.environ.get(k) for k in ("VPPX_VERT", "VPPX_V3_SPIN_LIMIT")}
    os.environ["VPPX_VERT"], os.environ["VPPX_V3_SPIN_LIMIT"] = "3", "1"
    try:
        ctx = _lib.Context()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    lib = _lib.load()
    b = synth.make_batch(B, H, W, D, 0.05, seed=21)
    lv, rv = np.empty_like(b["left"]), np.empty_like(b["right"])
    for f in range(B):
        oracle.init_rand(f)
        lv[f], rv[f] = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f])
    out = np.zeros((B, H, W), np.float32)
    p = _lib.rsgm_params(dmax=D)
    _lib.check(lib.vppx_rsgm_host(ctx.handle, C.byref(p), B, H, W, 3, _lib.np_ptr(b["left"]), _lib.np_ptr(lv), _lib.np_ptr(rv),
                                  None, None, _lib.np_ptr(out)))
    assert ctx.lockstep_failures == 1 and lib.vppx_uses_vert(ctx.handle) == 0
    for f in range(B):
        assert np.array_equal(oracle.compute_rsgm(b["left"][f], lv[f], rv[f], dmax=D), out[f]), f
    ctx.status()


def test_host_entry_point_does_not_swallow_an_earlier_calls_lost_lock_step():
    """An asynchronous fused call loses its lock step; the next call is the synchronous vppx_rsgm_host.  The mark belongs
    to the EARLIER launch, so it must be reported (once) instead of being consumed by the host call's own retry loop."""
    import torch
    from vppstereo_amd import _lib
    D = 192
    eng = _engine(VPPX_VERT=3, VPPX_V3_SPIN_LIMIT=1)
    b, args = _small_batch(eng, seed=14)
    lost = eng.vpp_rsgm(*args, seed=2, rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()              # the mark is there, nobody has asked yet
    lib, ctx = eng.lib, eng.ctx
    B, H, W = lost.shape
    lv, rv = np.empty_like(b["left"]), np.empty_like(b["right"])
    for f in range(B):
        oracle.init_rand(f)
        lv[f], rv[f] = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f])
    out = np.zeros((B, H, W), np.float32)
    p = _lib.rsgm_params(dmax=D)
    call = lambda: lib.vppx_rsgm_host(ctx.handle, C.byref(p), B, H, W, 3, _lib.np_ptr(b["left"]), _lib.np_ptr(lv), _lib.np_ptr(rv),
                                      None, None, _lib.np_ptr(out))
    assert call() == -8 and b"lost its lock step" in lib.vppx_last_error()
    assert ctx.lockstep_failures == 1 and bool(torch.isnan(lost).all())
    _lib.check(call())                    # reported once; the context has moved to the line-parallel kernel
    assert lib.vppx_uses_vert(ctx.handle) == 0
    assert np.array_equal(oracle.compute_rsgm(b["left"][0], lv[0], rv[0], dmax=D), out[0])


def test_forced_lockstep_timeout_under_graph_replay():
    """Graph mode: the captured graph holds the fused launch.  After a lost lock step the graph is dropped, the error is
    returned (not swallowed by a silent eager retry), and later calls capture the line-parallel layout."""
    import torch
    from vppstereo_amd._lib import VppxError
    D = 192
    ref = _engine(VPPX_VERT=0)
    eng = _engine(VPPX_VERT=3, VPPX_V3_SPIN_LIMIT=1)
    b, args = _small_batch(eng, seed=12)
    want = ref.vpp_rsgm(*args, seed=6, rsgm_kw=dict(dmax=D)).clone()
    ref.synchronize()
    out = torch.empty_like(want)
    side = torch.cuda.Stream(device=eng.device)
    with torch.cuda.stream(side):
        eng.set_graph_mode(True)
        seen = 0
        for _ in range(6):
            try:
                eng.vpp_rsgm(*args, out=out, seed=6, rsgm_kw=dict(dmax=D))
                eng.synchronize()
            except VppxError as e:
                assert "lost its lock step" in str(e)
                seen += 1
        assert seen == 1 and eng.ctx.lockstep_failures == 1
        assert eng.uses_vert() == 0 and eng.graph_replays() > 0
        assert torch.equal(out, want)


@pytest.mark.parametrize("p1", [62, 231, 232, 20000, 70000])
def test_large_p1_is_exact_on_both_layouts(p1):
    """P1 far above P2: the byte-volume kernels clamp it (exact, see rsgm_launch_paths); the oracle saturates in u16."""
    import torch
    D = 192
    fused, eight = _engine(VPPX_VERT=3), _engine(VPPX_VERT=0)
    b, args = _small_batch(fused, B=8, H=40, W=112, seed=30)
    kw = dict(dmax=D, p1=p1)
    of = fused.vpp_rsgm(*args, seed=1, rsgm_kw=kw)
    o8 = eight.vpp_rsgm(*args, seed=1, rsgm_kw=kw)
    fused.synchronize()
    eight.synchronize()
    assert fused.uses_vert() == 3 and eight.uses_vert() == 0
    assert torch.equal(of, o8)
    o1 = eight.vpp_rsgm(*[a[:1].contiguous() for a in args], seed=1, rsgm_kw=kw)   # 16 lanes x 12 variant
    eight.synchronize()
    assert torch.equal(o1[0], o8[0])
    oracle.init_rand(1)
    lo, ro = oracle.vpp(b["left"][0], b["right"][0], b["hints"][0])
    assert np.array_equal(oracle.compute_rsgm(b["left"][0], lo, ro, dmax=D, p1=p1), of[0].cpu().numpy())


def _torch_batches(eng, n, B=8, H=40, W=120, D=192):
    out = []
    for i in range(n):
        b = synth.make_batch(B, H, W, D, 0.05, seed=70 + i)
        out.append([_dev(eng, b[k]) for k in ("left", "right", "hints")])
    return out


def test_pipelining_is_safe_with_torch_produced_inputs_and_per_call_tensors():
    """Cross-call pipelining without any promise from the caller: every call's inputs are produced by torch kernels
    queued immediately before it, every input and output tensor is allocated per call and dropped right after (torch's
    allocator hands the memory to the next iteration), junk is written into fresh allocations in between.  The front
    stage must wait for the producers and must never write caller memory out of stream order."""
    import torch
    D = 192
    ref, eng = _engine(), _engine()
    eng.set_pipeline(True)
    base = _torch_batches(eng, 3)
    want, want_occ, want_lv = [], [], []
    for i, (l, r, h) in enumerate(base):
        occ = ref.occlusion_heuristic(h)
        lv = torch.empty_like(l)
        want.append(ref.vpp_rsgm(l, r, h, g_occ=occ, l_vpp=lv, seed=i, rsgm_kw=dict(dmax=D)).clone())
        want_occ.append(occ.clone())
        want_lv.append(lv)
    ref.synchronize()
    got = []
    for rep in range(4):
        for i, (bl, br, bh) in enumerate(base):
            l = bl.clone()                                  # producers: torch kernels on the current stream, just before the call
            r = torch.flip(torch.flip(br, [1]), [1])
            h = bh * 1.0
            occ = torch.empty(bh.shape, dtype=torch.uint8, device=eng.device)
            lv = torch.empty_like(l)
            out = eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occ, l_vpp=lv, seed=i, rsgm_kw=dict(dmax=D))
            got.append((i, out.double().sum(), occ.sum(), lv.sum(dtype=torch.int64), out))   # consumers right behind it
            del l, r, h, occ, lv
            junk = [torch.empty(bl.shape, dtype=torch.uint8, device=eng.device).fill_(7 + rep) for _ in range(3)]
            del junk
    eng.synchronize()
    for i, s_out, s_occ, s_lv, out in got:
        assert torch.equal(out, want[i]), i
        assert float(s_out) == float(want[i].double().sum()), i
        assert int(s_occ) == int(want_occ[i].sum()) and int(s_lv) == int(want_lv[i].sum(dtype=torch.int64)), i


def test_pipelining_with_an_inputs_ready_event_from_a_loader_stream():
    """The overlapped form: inputs are produced on a second ("loader") stream and the caller passes the event recorded
    there; the front stage waits for that event only.  The launch stream never waits for the loader explicitly: the
    library's own ordering (front stage -> aggregation) must cover it."""
    import torch
    D = 192
    ref, eng = _engine(), _engine()
    eng.set_pipeline(True)
    base = _torch_batches(eng, 3, B=12, H=33, W=70)
    want = []
    for i, (l, r, h) in enumerate(base):
        occ = ref.occlusion_heuristic(h)
        want.append(ref.vpp_rsgm(l, r, h, g_occ=occ, seed=i, rsgm_kw=dict(dmax=D)).clone())
    ref.synchronize()
    loader = torch.cuda.Stream(device=eng.device)
    keep, got = [], []
    torch.cuda.synchronize()
    for rep in range(4):
        for i, (bl, br, bh) in enumerate(base):
            with torch.cuda.stream(loader):
                l, r, h = bl.clone(), br.clone(), bh.clone()
                for _ in range(3):
                    h = h * 1.0                              # a loader with some latency
                ev = torch.cuda.Event()
                ev.record(loader)
            out = eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", seed=i, rsgm_kw=dict(dmax=D), inputs_ready=ev)
            got.append((i, out))
            keep.append((l, r, h, ev))                       # loader-stream tensors stay alive until the end
    eng.synchronize()
    for i, out in got:
        assert torch.equal(out, want[i]), i


def test_fused_kernels_soak_alone_and_next_to_competing_work():
    """The lock-step hand-off under load: 60 back-to-back fused steps must all give the first step's bits, alone and with
    bf16 GEMMs of a second stream competing for the CUs (which may slow the launches down or, at worst, trip the bounded
    waits -- that would raise -- but never change a result silently).  Both fused kernels (8 and 16 pixels per wave)."""
    import torch
    D = 192
    for ppw in (16, 8):
        eng = _engine(VPPX_VERT=3, VPPX_V3_PPW=ppw)
        b, args = _small_batch(eng, B=16, H=270, W=480, seed=40 + ppw)
        ref = eng.vpp_rsgm(*args, seed=2, rsgm_kw=dict(dmax=D)).clone()
        eng.synchronize()
        assert eng.uses_vert() == 3 and eng.fused_pixels_per_wave() == ppw
        out = torch.empty_like(ref)
        for i in range(60):
            eng.vpp_rsgm(*args, out=out, seed=2, rsgm_kw=dict(dmax=D))
            if i % 6 == 5:
                eng.synchronize()
                assert torch.equal(out, ref), (ppw, i)
        side = torch.cuda.Stream(device=eng.device)
        a = torch.randn(4096, 4096, device=eng.device, dtype=torch.bfloat16)
        for i in range(24):
            with torch.cuda.stream(side):
                for _ in range(3):
                    a @ a
            eng.vpp_rsgm(*args, out=out, seed=2, rsgm_kw=dict(dmax=D))
            if i % 6 == 5:
                eng.synchronize()
                assert torch.equal(out, ref), (ppw, i)
        torch.cuda.synchronize()
        eng.status()
        assert eng.uses_vert() == 3


def test_batch_quantum_is_a_whole_round_of_the_lock_step_kernel():
    """vppx_batch_quantum: frames per full round of (frame, pass) groups of the 16-pixels-per-wave kernel.  A hint, not a
    constraint: batches of one quantum, and of a quantum + 1 (a part-filled round, a ghost round of groups), both take the
    fused layout and give the 8-path layout's bits."""
    import torch
    eng = _engine()
    q = eng.batch_quantum(60, 480, 192)
    assert q > 0 and q % 4 == 0          # whole groups per XCD x 8 XCDs / 2 passes
    assert eng.batch_quantum(540, 960, 192) > 0 and eng.batch_quantum(375, 1242, 192) > 0
    assert eng.batch_quantum(375, 1242, 192) <= eng.batch_quantum(540, 960, 192)   # wider frames: fewer groups resident
    assert eng.batch_quantum(60, 480, 100) == 0                                   # no fused layout for this range
    ref_eng = _engine(VPPX_VERT=0)
    for B in (8, 9):
        b, args = _small_batch(eng, B=B, H=60, W=480, seed=70 + B)
        out = eng.vpp_rsgm(*args, seed=3, rsgm_kw=dict(dmax=192))
        eng.synchronize()
        assert eng.uses_vert() == 3
        ref = ref_eng.vpp_rsgm(*[a.to(ref_eng.device) for a in args], seed=3, rsgm_kw=dict(dmax=192))
        ref_eng.synchronize()
        assert torch.equal(out, ref), B


def test_pipelined_stream_of_small_batches_starts_the_next_front_stage_early():
    """Fewer than 8 frames per call: the next call's front stage starts next to THIS call's aggregation (two alternating
    sets of gray / census images).  A stream that mixes single frames, small and large batches, with no synchronisation
    in between, must give the unpipelined results call by call."""
    import torch
    ref_eng, eng = _engine(), _engine()
    eng.set_pipeline(True)
    dev = eng.device
    shapes = [(1, 60, 200), (1, 60, 200), (2, 45, 130), (8, 40, 120), (1, 60, 200), (3, 33, 90), (12, 40, 120), (1, 45, 130), (1, 60, 200)]
    batches, outs, occs, refs = [], [], [], []
    for i, (B, H, W) in enumerate(shapes):
        b = synth.make_batch(B, H, W, 192, 0.05, seed=300 + i)
        batches.append([_dev(eng, b[k]) for k in ("left", "right", "hints")])
        outs.append([torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(3)])
        occs.append(torch.empty((B, H, W), dtype=torch.uint8, device=dev))
    for i, (l, r, h) in enumerate(batches):
        refs.append(ref_eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", seed=i, rsgm_kw=dict(dmax=192)).clone())
    ref_eng.synchronize()
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    ev.record()
    torch.cuda.synchronize()
    for rep in range(3):
        for i, (l, r, h) in enumerate(batches):
            eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occs[i], out=outs[i][rep], seed=i, rsgm_kw=dict(dmax=192),
                         inputs_ready=ev)
    eng.synchronize()
    for rep in range(3):
        for i in range(len(batches)):
            assert torch.equal(outs[i][rep], refs[i]), (rep, i, shapes[i])


@pytest.mark.parametrize("B", [9, 20, 26])
def test_full_size_odd_batches_take_the_fused_layout(B):
    """Any number of frames from 8 on takes the fused layout at full size (the grid is rounded up to whole rounds of 8
    groups, ghost groups exit): 9, 20 and 26 frames of 540 x 960 x 192, two pipelined steps, equal the 8-path layout bit for bit.
    9 frames and the 10-frame part of 26 are under-filled lock-step launches (W/E runs next to them on the side stream); 20
    frames are one part of a whole round + 4: two lock-step launches, W/E next to the second (rsgm_vert3_plan)."""
    import torch
    H, W, D = 540, 960, 192
    eng = _engine()
    eng.set_pipeline(True)
    ref = _engine(VPPX_VERT=0)
    nu = 4
    b = synth.make_batch(nu, H, W, D, 0.03, seed=99 + B)
    idx = [(i * 3) % nu for i in range(B)]
    left, right, hints = (_dev(eng, np.ascontiguousarray(b[k][idx])) for k in ("left", "right", "hints"))
    outs = [torch.empty((B, H, W), dtype=torch.float32, device=eng.device) for _ in range(2)]
    for s in range(2):
        eng.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", out=outs[s], seed=5 + s, rsgm_kw=dict(dmax=D))
    eng.synchronize()
    assert eng.uses_vert() == 3
    for s in range(2):
        want = ref.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", seed=5 + s, rsgm_kw=dict(dmax=D))
        ref.synchronize()
        assert ref.uses_vert() == 0
        assert torch.equal(outs[s], want), (B, s)


# (B, chunk, parts): the remainder goes first, and below 8 frames it rides on the first full part: 13 = (3 + 5) + 5, 16 = 8 + 8,
# 9 = (1 + 4) + 4, 27 = (3 + 8) + 8 + 8, 30 = (6 + 12) + 12 and 20 = 8 + 12 (uneven parts of 8 and more frames: fused layout)
@pytest.mark.parametrize("B,chunk,parts", [(13, 5, 2), (16, 8, 2), (9, 4, 2), (27, 8, 3), (30, 12, 2), (20, 12, 2)])
def test_parts_of_a_split_batch_equal_the_unsplit_call(B, chunk, parts):
    """A batch larger than one round of the lock-step kernel runs as consecutive parts (vppx_api.hip: vpp_rsgm_parts).  Forced
    here with VPPX_CHUNK on small frames: every output of the call -- disparities, mask, patterned pair -- must equal the
    unsplit call's (frame f draws from srand(seed + f) whatever the split), with and without the cross-call overlap; the
    split must really have happened (vppx_last_call_parts), parts of growing size re-allocate the workspace while pipelined."""
    import torch
    H, W, D = 48, 112, 192
    split, whole = _engine(VPPX_CHUNK=chunk), _engine(VPPX_CHUNK=0)
    b = synth.make_batch(B, H, W, D, 0.05, seed=77)
    args = [_dev(split, b[k]) for k in ("left", "right", "hints")]
    outs = {}
    for name, eng in (("split", split), ("whole", whole)):
        for piped in (False, True):
            eng.set_pipeline(piped)
            occ = torch.empty((B, H, W), dtype=torch.uint8, device=eng.device)
            lv = torch.empty((B, H, W, 3), dtype=torch.uint8, device=eng.device)
            rv = torch.empty_like(lv)
            o = eng.vpp_rsgm(*args, g_occ="occlusion_heuristic", occ_out=occ, l_vpp=lv, r_vpp=rv, seed=5, rsgm_kw=dict(dmax=D, subpixel=1))
            eng.synchronize()
            assert eng.last_call_parts() == (parts if name == "split" else 1), (name, eng.last_call_parts())
            if B >= 16:
                assert eng.uses_vert() == 3
            outs[(name, piped)] = (o.clone(), occ, lv, rv)
    ref = outs[("whole", False)]
    for key, got in outs.items():
        for a, w in zip(got, ref):
            assert torch.equal(a, w), key
    f = B - 1   # the last frame sits in the last part: its seed is 5 + f
    conf = oracle.occlusion_heuristic(b["hints"][f])[1]
    oracle.init_rand(5 + f)
    lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f], g_occ=conf)
    assert np.array_equal(lo, ref[2][f].cpu().numpy()) and np.array_equal(ro, ref[3][f].cpu().numpy())
    assert np.array_equal(oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D, subpixel=True), ref[0][f].cpu().numpy())
