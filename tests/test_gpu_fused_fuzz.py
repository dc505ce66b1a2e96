"""GPU: the fused aggregation layout (W/E line-parallel + sgm_vert3_kernel) against the 8-path layout
on random shapes, batch sizes and penalty parameters.  The two layouts share no aggregation code for the six vertical
and diagonal paths (register-resident lock-step kernel with neighbour hand-off vs one independent chain per scan line),
so equality of every disparity is a strong check; a few frames are also compared with the CPU oracle."""
import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    import os
    import torch
    from vppstereo_amd.engine import Engine
    assert torch.cuda.is_available()
    old = os.environ.get("VPPX_VERT")
    try:
        os.environ["VPPX_VERT"] = "3"   # fused whenever the shape allows it
        os.environ["VPPX_V3_PPW"] = "8"   # ... with the 8-pixels-per-wave kernel at every D
        fused = Engine()
        os.environ["VPPX_V3_PPW"] = "16"  # ... with the 16-pixels-per-wave kernel (whatever the batch size)
        wide = Engine()
        os.environ.pop("VPPX_V3_PPW")
        os.environ["VPPX_VERT"] = "0"   # always the eight line-parallel paths
        eight = Engine()
    finally:
        os.environ.pop("VPPX_V3_PPW", None)
        if old is None:
            os.environ.pop("VPPX_VERT", None)
        else:
            os.environ["VPPX_VERT"] = old
    return fused, eight, wide


def _case(rng):
    B = int(rng.choice([4, 8, 9, 11, 12, 13, 16]))  # any batch size: the fused grid is rounded up to whole rounds of 8 groups
    H = int(rng.integers(5, 70))
    W = int(rng.integers(5, 330))
    p2min = int(rng.integers(5, 40))
    kw = dict(dmax=int(rng.choice([64, 128, 192, 192, 256])), p1=int(rng.integers(1, 25)), p2min=p2min, gamma=int(rng.integers(p2min, 62)),
              alpha=float(rng.choice([0.0, 0.25, 0.5, 1.0])), subpixel=int(rng.integers(0, 2)),
              uniqueness=float(rng.choice([0.95, 0.8, 1.0])))
    return B, H, W, kw


@pytest.mark.parametrize("seed", range(24))
def test_fused_layout_equals_eight_path_layout(engines, seed):
    import torch
    fused, eight, wide = engines
    rng = np.random.default_rng(1000 + seed)
    B, H, W, kw = _case(rng)
    if seed == 0:
        B, H, W = 8, 5, 5            # the smallest frame the library takes: pads to 16 x 16, two 8-column waves, both at a border
    if seed == 1:
        kw.update(p2min=61, gamma=61, alpha=0.0)   # 3 * (24 + 61) = 255: the largest sums a byte volume can hold
    b = synth.make_batch(B, H, W, kw["dmax"], float(rng.choice([0.0, 0.03, 0.3])), seed=seed)
    dev = fused.device
    args = [torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("left", "right", "hints")]
    out_f = fused.vpp_rsgm(*args, seed=seed, rsgm_kw=kw)
    out_8 = eight.vpp_rsgm(*args, seed=seed, rsgm_kw=kw)
    torch.cuda.synchronize()
    assert fused.uses_vert() == 3 and eight.uses_vert() == 0, (B, H, W, kw)
    assert torch.equal(out_f, out_8), (B, H, W, kw, int((out_f != out_8).sum()))
    # the 16-pixels-per-wave kernel (sgm_vert4_kernel, every D; the default once a batch fills the chip with its groups)
    out_w = wide.vpp_rsgm(*args, seed=seed, rsgm_kw=kw)
    wide.synchronize()
    assert wide.uses_vert() == 3
    assert torch.equal(out_w, out_8), (B, H, W, kw, int((out_w != out_8).sum()))
    if seed % 5 == 0:   # and against the oracle (frame 0 and the last one)
        got = out_f.cpu().numpy()
        for f in (0, B - 1):
            oracle.init_rand(seed + f)
            lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f])
            okw = dict(kw)
            okw["subpixel"] = bool(okw["subpixel"])
            assert np.array_equal(oracle.compute_rsgm(b["left"][f], lo, ro, **okw), got[f]), (f, B, H, W, kw)


def test_fused_layout_wants_byte_sized_three_path_sums(engines):
    """3 * (24 + P2max) > 255: the three-path sum does not fit a byte, the library must stay on the 8-path layout."""
    import torch
    fused, _, _ = engines
    b = synth.make_batch(8, 20, 40, 192, 0.05, seed=3)
    args = [torch.from_numpy(np.ascontiguousarray(b[k])).to(fused.device) for k in ("left", "right", "hints")]
    out = fused.vpp_rsgm(*args, seed=1, rsgm_kw=dict(dmax=192, p2min=17, gamma=62, alpha=0.5))
    torch.cuda.synchronize()
    assert fused.uses_vert() == 0
    oracle.init_rand(1)
    lo, ro = oracle.vpp(b["left"][0], b["right"][0], b["hints"][0])
    assert np.array_equal(oracle.compute_rsgm(b["left"][0], lo, ro, dmax=192, p2min=17, gamma=62, alpha=0.5), out[0].cpu().numpy())


def test_cross_call_pipelining_keeps_results_and_stream_order():
    """Engine.set_pipeline: the front stage of call k+1 runs on a second stream under call k's sum / post kernels.  A
    stream of different batches (with the occlusion mask) must give the unpipelined results, and torch work queued on
    the current stream right after a call must see that call's finished disparities.  The buffers the front stage
    writes (the occlusion mask) are persistent, as the mode requires: a tensor allocated per call could reuse memory
    that work still queued on the main stream has not read yet (torch's allocator only knows the main stream)."""
    import torch
    from vppstereo_amd.engine import Engine
    ref_eng, eng = Engine(), Engine()
    eng.set_pipeline(True)
    dev = eng.device
    shapes = [(8, 40, 120), (8, 40, 120), (12, 33, 70), (4, 25, 50), (8, 40, 120)]
    batches, occ_bufs, out_bufs = [], [], []
    for i, (B, H, W) in enumerate(shapes):
        b = synth.make_batch(B, H, W, 192, 0.05, seed=50 + i)
        batches.append([torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("left", "right", "hints")])
        occ_bufs.append(torch.empty((B, H, W), dtype=torch.uint8, device=dev))
        out_bufs.append([torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(3)])
    refs = []
    for i, (l, r, h) in enumerate(batches):
        occ = ref_eng.occlusion_heuristic(h)
        refs.append(ref_eng.vpp_rsgm(l, r, h, g_occ=occ, seed=i, rsgm_kw=dict(dmax=192)).clone())
    torch.cuda.synchronize()
    sums = []
    for rep in range(3):          # no synchronisation inside: calls, and torch reductions right behind them, pile up
        for i, (l, r, h) in enumerate(batches):
            occ = eng.occlusion_heuristic(h, out=occ_bufs[i])
            out = eng.vpp_rsgm(l, r, h, g_occ=occ, out=out_bufs[i][rep], seed=i, rsgm_kw=dict(dmax=192))
            sums.append(out.double().sum())       # torch kernel on the same stream, queued immediately
    torch.cuda.synchronize()
    for rep in range(3):
        for i in range(len(batches)):
            assert torch.equal(out_bufs[i][rep], refs[i]), (rep, i)
            assert float(sums[rep * len(batches) + i]) == float(refs[i].double().sum()), (rep, i)
    # other entry points between pipelined calls use the same workspace on the launch stream: the next front stage has
    # to wait for them (the library drops its "previous aggregation done" shortcut on any other call)
    l0, r0, h0 = batches[0]
    l2, r2, h2 = batches[2]
    ref_lv, ref_rv = ref_eng.vpp(l2, r2, h2, seed=9)
    ref_sep = ref_eng.rsgm(l2, ref_lv, ref_rv, dmax=192).clone()
    torch.cuda.synchronize()
    for rep in range(3):
        occ = eng.occlusion_heuristic(h0, out=occ_bufs[0])
        a = eng.vpp_rsgm(l0, r0, h0, g_occ=occ, out=out_bufs[0][0], seed=0, rsgm_kw=dict(dmax=192))
        lv, rv = eng.vpp(l2, r2, h2, seed=9)                     # not pipelined: VPP workspace, launch stream
        sep = eng.rsgm(l2, lv, rv, out=out_bufs[2][0], dmax=192)  # not pipelined: gray / census / volumes, launch stream
        occ = eng.occlusion_heuristic(h0, out=occ_bufs[0])
        b = eng.vpp_rsgm(l0, r0, h0, g_occ=occ, out=out_bufs[0][1], seed=0, rsgm_kw=dict(dmax=192))
        torch.cuda.synchronize()
        assert torch.equal(a, refs[0]) and torch.equal(b, refs[0]) and torch.equal(sep, ref_sep), rep
        assert torch.equal(lv, ref_lv) and torch.equal(rv, ref_rv), rep


@pytest.mark.parametrize("seed", range(16))
def test_batched_vpp_with_a_mask_random_parameters(engines, seed):
    """The fused entry point with 8+ frames and an occlusion mask takes the two-pass L side (pixels whose window holds an
    occluded hint are deferred to a work list) and the early first pass next to the R list build: patterned pairs against the
    oracle's vpp() frame by frame, over patch sizes, colour modes, blending weights, scan directions and mask densities."""
    import torch
    fused = engines[0]
    rng = np.random.default_rng(7000 + seed)
    B = int(rng.choice([8, 9, 12]))
    H, W = int(rng.integers(8, 48)), int(rng.integers(12, 130))
    dens = float(rng.choice([0.02, 0.1, 0.4]))
    b = synth.make_batch(B, H, W, 64, dens, seed=7000 + seed)
    occ = ((rng.random((B, H, W)) < float(rng.choice([0.1, 0.5, 1.0]))) & (b["hints"] > 0)).astype(np.uint8)
    wsize = int(rng.choice([1, 3, 5, 7, 9]))
    ref_kw = dict(wsize=wsize, left2right=bool(rng.integers(2)), blending=float(rng.choice([0.4, 0.0, 1.0, 0.73])),
                  uniform_color=bool(rng.integers(2)), c_occ=float(rng.choice([0.0, 0.2, 1.0])), discard_occ=bool(rng.integers(4) == 0),
                  interpolate=bool(rng.integers(2)))
    kw = dict(wsize=wsize, direction=int(ref_kw["left2right"]), c=ref_kw["blending"], uniform_color=int(ref_kw["uniform_color"]),
              c_occ=ref_kw["c_occ"], discard_occluded=int(ref_kw["discard_occ"]), interpolate=int(ref_kw["interpolate"]))
    dev = fused.device
    args = [torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("left", "right", "hints")]
    lv = torch.empty_like(args[0])
    rv = torch.empty_like(args[0])
    fused.vpp_rsgm(*args, g_occ=torch.from_numpy(occ).to(dev), l_vpp=lv, r_vpp=rv, seed=seed, vpp_kw=kw, rsgm_kw=dict(dmax=64))
    fused.synchronize()
    lv, rv = lv.cpu().numpy(), rv.cpu().numpy()
    for f in range(B):
        oracle.init_rand(seed + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f], g_occ=occ[f], **ref_kw)
        assert np.array_equal(lo, lv[f]) and np.array_equal(ro, rv[f]), (seed, f, B, H, W, ref_kw)


def _env_engine(**env):
    import os
    from vppstereo_amd.engine import Engine
    old = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update({k: str(v) for k, v in env.items()})
        return Engine()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# (engine environment, B, H, W, occlusion parameters): what the front stage of a PIPELINED call at dmax = 256 picks when the
# previous call's sum / WTA kernel leaves little LDS (vpp_rsgm_one: front_lds_budget; vpp_kernels.hip: occ_test_kernel<NJ, 2>,
# apply_l_bits_kernel<NW, 1>, r_rows_kernel in column segments).  Fewer than 6 frames: 8-path layout, uniform ring of 153 KB,
# 6.5 KB left -> the small variants, rows of 450 / 333 / 211 columns = 3 / 2 / 2 segments of unequal width; 8 frames with
# sum_trap1 / sum_trap0 (trapezoid ring without spare slots: 42 KB left, uniform ring: 6.5 KB): the fused layout's budgets;
# the default (spare slots: nothing fits next to the kernel, the front stage keeps its stand-alone shapes).
@pytest.mark.parametrize("env,B,H,W,occ", [
    (dict(), 3, 20, 450, dict()),
    (dict(), 5, 14, 333, dict(rx=17, ry=13, th_filter=0.5)),       # 238 window positions: occ_test_kernel<4, 2>
    (dict(), 2, 18, 211, dict(rx=21, ry=19, l=1.0, g=0.25)),        # 420: the generic variant
    (dict(VPPX_VERT=3, VPPX_VARIANT="sum_trap1"), 8, 10, 1500, dict()),
    (dict(VPPX_VERT=3, VPPX_VARIANT="sum_trap0"), 8, 12, 420, dict(rx=17, ry=13)),
    (dict(VPPX_VERT=3), 8, 12, 300, dict()),
])
def test_pipelined_front_stage_at_d256_matches_unpipelined_and_oracle(env, B, H, W, occ):
    """Three back-to-back pipelined calls at dmax = 256 with the mask computed on the way: disparities, mask and patterned
    pair equal the unpipelined engine bit for bit, and frame 0 / the last frame equal the CPU oracle."""
    import torch
    D = 256
    eng, ref = _env_engine(**env), _env_engine(**env)
    eng.set_pipeline(True)
    dev = eng.device
    b = synth.make_batch(B, H, W, D, 0.06, seed=9100 + W)
    args = [torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("left", "right", "hints")]
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    ev.record()
    torch.cuda.synchronize()
    got = []
    for s in range(3):   # no synchronisation in between: the front stage of call s + 1 runs under call s's sum / WTA kernel
        occ_o = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        lv = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
        rv = torch.empty_like(lv)
        o = eng.vpp_rsgm(*args, g_occ=dict(occ) if occ else "occlusion_heuristic", occ_out=occ_o, l_vpp=lv, r_vpp=rv, seed=40 + s,
                         rsgm_kw=dict(dmax=D, subpixel=1), inputs_ready=ev)
        got.append((o, occ_o, lv, rv))
    eng.synchronize()
    if "VPPX_VERT" in env:
        assert eng.uses_vert() == 3
    for s in range(3):
        occ_o = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        lv = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
        rv = torch.empty_like(lv)
        o = ref.vpp_rsgm(*args, g_occ=dict(occ) if occ else "occlusion_heuristic", occ_out=occ_o, l_vpp=lv, r_vpp=rv, seed=40 + s,
                         rsgm_kw=dict(dmax=D, subpixel=1))
        ref.synchronize()
        for a, w, name in zip(got[s], (o, occ_o, lv, rv), ("disparity", "mask", "l_vpp", "r_vpp")):
            assert torch.equal(a, w), (s, name)
    for f in (0, B - 1):
        conf = oracle.occlusion_heuristic(b["hints"][f], **occ)[1]
        assert np.array_equal(conf, got[2][1][f].cpu().numpy()), f
        oracle.init_rand(42 + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f], g_occ=conf)
        assert np.array_equal(lo, got[2][2][f].cpu().numpy()) and np.array_equal(ro, got[2][3][f].cpu().numpy()), f
        assert np.array_equal(oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D, subpixel=True), got[2][0][f].cpu().numpy()), f


@pytest.mark.parametrize("B,H,W", [(8, 9, 5), (8, 12, 44), (9, 7, 64), (8, 10, 70), (6, 21, 130), (8, 6, 200), (3, 17, 333), (8, 5, 1030)])
def test_trapezoid_ring_at_d256_every_width(engines, B, H, W):
    """sum_wta_trap_kernel (D = 256, fused layout): padded widths below, at and above one 64-pixel round, rows that end inside a
    round, blocks of one row and of many, 3 to 9 frames -- against the 8-path layout (uniform ring, 32-pixel rounds, another
    kernel) bit for bit, frame 0 against the oracle; with and without sub-pixel refinement."""
    import torch
    fused, eight, wide = engines
    b = synth.make_batch(B, H, W, 256, 0.06, seed=8800 + W)
    args = [torch.from_numpy(np.ascontiguousarray(b[k])).to(fused.device) for k in ("left", "right", "hints")]
    for sub in (1, 0):
        kw = dict(dmax=256, subpixel=sub, p1=7, p2min=13, gamma=40)
        a = fused.vpp_rsgm(*args, seed=3, rsgm_kw=kw)
        fused.synchronize()
        assert fused.uses_vert() == 3
        w = eight.vpp_rsgm(*args, seed=3, rsgm_kw=kw)
        eight.synchronize()
        assert eight.uses_vert() == 0
        assert torch.equal(a, w), (B, H, W, sub)
        c = wide.vpp_rsgm(*args, seed=3, rsgm_kw=kw)
        wide.synchronize()
        assert torch.equal(c, w), (B, H, W, sub)
    oracle.init_rand(3)
    lo, ro = oracle.vpp(b["left"][0], b["right"][0], b["hints"][0])
    assert np.array_equal(oracle.compute_rsgm(b["left"][0], lo, ro, dmax=256, subpixel=False, p1=7, p2min=13, gamma=40), a[0].cpu().numpy())


# (D, B, H, W, layer): `we_layer=N` makes the W/E launcher take a layer of waves to be N (the device's SIMD count on a real launch), so
# that small launches have a part-filled last layer whose lines are cut into pieces: 2 x B x Hp / 4 line quads = whole layers + a tail;
# c pieces per tail line, c in 2..8 with the smallest ceil(tail * c / layer) / c, if that is below 0.9 (launch_we12).
_SPLIT_CASES = [
    (192, 8, 20, 96, 60),      # 128 line quads = 2 x 60 + 8: 2 pieces of 4 groups, both boundaries inside the d > x columns
    (192, 8, 20, 330, 90),     # 128 = 90 + 38: 7 pieces of 4 groups over 28 (load 3 / 7)
    (192, 9, 37, 200, 100),    # 216 = 2 x 100 + 16: 2 pieces over 18 groups, the last of them a part of a group (208 columns)
    (128, 8, 33, 150, 89),     # D = 128 (8 steps per group): 192 = 2 x 89 + 14: 2 pieces over 20 groups
    (256, 8, 12, 420, 61),     # D = 256 (16 steps per group): 64 = 61 + 3: 2 pieces over 27 groups
    (192, 16, 60, 130, 200),   # 512 = 2 x 200 + 112: 7 pieces (load 4 / 7) over 12 groups: one group each, the last takes six
    (192, 12, 28, 160, 97),    # 192 = 97 + 95: a tail of nearly a whole layer -- nothing is cut (same results either way)
]


@pytest.mark.parametrize("D,B,H,W,layer", _SPLIT_CASES)
@pytest.mark.parametrize("lq", ["we_lq0", "we_lq1"])
def test_we_lines_cut_into_pieces_equal_whole_lines(D, B, H, W, layer, lq):
    """The W/E kernel cuts the lines of its last, part-filled layer of waves into pieces that hand the path state on through memory
    (rsgm_kernels.hip, We12Args): the volumes -- and so every disparity -- must equal those of whole lines (`we_whole`) and of the
    8-path layout, frame 0 the oracle's; and when no piece ever publishes (`we_mute`) every later piece gives up and computes its
    line from the start: same values again."""
    import torch
    cut = _env_engine(VPPX_VERT=3, VPPX_VARIANT=f"{lq},we_layer={layer}")
    whole = _env_engine(VPPX_VERT=3, VPPX_VARIANT=f"{lq},we_whole")
    mute = _env_engine(VPPX_VERT=3, VPPX_VARIANT=f"{lq},we_layer={layer},we_mute")
    eight = _env_engine(VPPX_VERT=0)
    b = synth.make_batch(B, H, W, D, 0.05, seed=7700 + W)
    args = [torch.from_numpy(np.ascontiguousarray(b[k])).to(cut.device) for k in ("left", "right", "hints")]
    kw = dict(dmax=D, p1=9, p2min=14, gamma=41, alpha=0.5)
    outs = []
    for eng in (cut, whole, mute, eight):
        for _ in range(2):           # twice: the second launch meets the first one's flags (another serial)
            o = eng.vpp_rsgm(*args, seed=5, rsgm_kw=kw)
            eng.synchronize()
        outs.append(o)
    assert cut.uses_vert() == 3 and whole.uses_vert() == 3 and mute.uses_vert() == 3 and eight.uses_vert() == 0
    for o, name in zip(outs[:3], ("cut", "whole", "mute")):
        assert torch.equal(o, outs[3]), (name, int((o != outs[3]).sum()))
    oracle.init_rand(5)
    lo, ro = oracle.vpp(b["left"][0], b["right"][0], b["hints"][0])
    assert np.array_equal(oracle.compute_rsgm(b["left"][0], lo, ro, **kw), outs[0][0].cpu().numpy())
