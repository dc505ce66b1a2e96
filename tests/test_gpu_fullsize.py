"""GPU parity at the FULL sizes of BASELINE.json's configurations (one frame each; the CPU oracle needs seconds
per frame), the reference's own rsgm.py call sequence through the drop-in natives, and the robustness items of
round 2 (stream ordering on torch's default stream, one rand stream state, argument checks)."""
import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4  # float disparity tolerance stated by BASELINE.json north_star (in practice the maps are bit-equal)


@pytest.fixture(scope="module")
def eng():
    import torch
    from vppstereo_amd.engine import Engine
    assert torch.cuda.is_available()
    return Engine()


def _dev(eng, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


def _fused_vs_oracle(eng, H, W, D, p, seed, with_occ=True):
    import torch
    fr = synth.make_frame(H, W, D, p, seed=seed)
    hints = _dev(eng, fr["hints"][None])
    occ = eng.occlusion_heuristic(hints) if with_occ else None
    lv = torch.empty((1, H, W, 3), dtype=torch.uint8, device=eng.device)
    rv = torch.empty_like(lv)
    out = eng.vpp_rsgm(_dev(eng, fr["left"][None]), _dev(eng, fr["right"][None]), hints, g_occ=occ, l_vpp=lv, r_vpp=rv,
                       seed=77, rsgm_kw=dict(dmax=D, subpixel=1))
    torch.cuda.synchronize()
    conf = oracle.occlusion_heuristic(fr["hints"])[1] if with_occ else None
    if with_occ:
        assert np.array_equal(conf, occ[0].cpu().numpy())
    oracle.init_rand(77)
    lo, ro = oracle.vpp(fr["left"], fr["right"], fr["hints"], g_occ=conf)
    assert np.array_equal(lo, lv[0].cpu().numpy()) and np.array_equal(ro, rv[0].cpu().numpy()), "pattern grid differs"
    want = oracle.compute_rsgm(fr["left"], lo, ro, dmax=D, subpixel=True)
    got = out[0].cpu().numpy()
    assert np.max(np.abs(want - got)) <= TOL, float(np.max(np.abs(want - got)))
    assert np.array_equal(want, got)
    return fr, got


def test_cfg1_540x960_d64_full_size(eng):
    """configs[0] (the reference's CPU-runnable case) on the HIP path: 540x960, 3 % hints, D = 64."""
    _fused_vs_oracle(eng, 540, 960, 64, 0.03, seed=101)


def test_cfg2_540x960_d192_full_size_with_occlusion_mask(eng):
    """configs[1], the benchmarked workload: occlusion heuristic + VPP + rSGM, D = 192."""
    fr, got = _fused_vs_oracle(eng, 540, 960, 192, 0.03, seed=102)
    assert np.median(np.abs(got - fr["gt"])) < 2.0


def test_cfg3_kitti_375x1242_d192_full_size(eng):
    """configs[2]: KITTI-sized frame (pads to 384x1248), 5 % hints, D = 192."""
    _fused_vs_oracle(eng, 375, 1242, 192, 0.05, seed=103)


@pytest.mark.parametrize("H,W,p,n_oracle", [(540, 960, 0.03, 2), (375, 1242, 0.05, 1)])
def test_batched_fused_vertical_layout_full_size(eng, H, W, p, n_oracle, monkeypatch):
    """Batches of 8+ frames aggregate N/NW/NE and S/SW/SE with the fused lock-step kernel (sgm_vert3_kernel; one frame
    per call, as in the tests above, takes the eight line-parallel paths).  At full size: every frame of a batch of 8
    equals the 8-path layout bit for bit, and the first `n_oracle` frames equal the CPU oracle."""
    import torch
    from vppstereo_amd.engine import Engine
    B, D = 8, 192
    b = synth.make_batch(B, H, W, D, p, seed=300 + H)
    args = [_dev(eng, b[k]) for k in ("left", "right", "hints")]
    out = eng.vpp_rsgm(*args, seed=5, rsgm_kw=dict(dmax=D, subpixel=1))
    torch.cuda.synchronize()
    assert eng.uses_vert() == 3
    monkeypatch.setenv("VPPX_VERT", "0")
    eng8 = Engine()
    out8 = eng8.vpp_rsgm(*args, seed=5, rsgm_kw=dict(dmax=D, subpixel=1))
    torch.cuda.synchronize()
    assert eng8.uses_vert() == 0
    assert torch.equal(out, out8)
    got = out.cpu().numpy()
    for f in range(n_oracle):
        oracle.init_rand(5 + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f])
        assert np.array_equal(oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D, subpixel=True), got[f]), f


def test_cfg5_batch8_fused_wide_kernel_full_size(monkeypatch):
    """configs[4] as bench.py's other_configs times it: 8 frames of 1536x2048, D = 256, the mask computed in the call.  The
    default layout is the fused one with the 16-pixels-per-wave kernel (32-block groups, two per XCD): every frame equals
    the 8-path layout bit for bit; one frame is also compared with the CPU oracle."""
    import torch
    from vppstereo_amd.engine import Engine
    B, H, W, D = 8, 1536, 2048, 256
    b = synth.make_batch(B, H, W, D, 0.01, seed=555)
    eng = Engine()
    args = [torch.from_numpy(np.ascontiguousarray(b[k])).to(eng.device) for k in ("left", "right", "hints")]
    occ = torch.empty((B, H, W), dtype=torch.uint8, device=eng.device)
    out = eng.vpp_rsgm(*args, g_occ="occlusion_heuristic", occ_out=occ, seed=8, rsgm_kw=dict(dmax=D, subpixel=1))
    eng.synchronize()
    assert eng.uses_vert() == 3 and eng.fused_pixels_per_wave() == 16
    monkeypatch.setenv("VPPX_VERT", "0")
    eng8 = Engine()
    out8 = eng8.vpp_rsgm(*args, g_occ="occlusion_heuristic", seed=8, rsgm_kw=dict(dmax=D, subpixel=1))
    eng8.synchronize()
    assert eng8.uses_vert() == 0
    bad = (out != out8).flatten(1).any(1).nonzero().flatten().tolist()
    assert not bad, bad
    f = 5
    conf = oracle.occlusion_heuristic(b["hints"][f])[1]
    assert np.array_equal(conf, occ[f].cpu().numpy())
    oracle.init_rand(8 + f)
    lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f], g_occ=conf)
    assert np.array_equal(oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D, subpixel=True), out[f].cpu().numpy())


def test_cfg5_1536x2048_d256_full_size(eng):
    """configs[4]: 1536x2048, 1 % hints, D = 256 (0.8 G cells; ~40 s of oracle time)."""
    _fused_vs_oracle(eng, 1536, 2048, 256, 0.01, seed=105, with_occ=False)


def test_cfg4_540x960_vpp_to_bf16_nchw_to_psmnet_volume_full_size(eng):
    """configs[3]: VPP -> bf16 NCHW network tensors (test.py:179-200) -> PSMNet concat volume + hint modulation
    (psmnet.py:157-197) at 540x960 / maxdisp 192, device resident end to end."""
    import torch
    import torch.nn.functional as F
    from oracle import frontends as FO
    H, W, D = 540, 960, 192
    fr = synth.make_frame(H, W, D, 0.03, seed=104)
    hints = _dev(eng, fr["hints"][None])
    lv, rv = eng.vpp(_dev(eng, fr["left"][None]), _dev(eng, fr["right"][None]), hints, seed=9)
    oracle.init_rand(9)
    lo, ro = oracle.vpp(fr["left"], fr["right"], fr["hints"])
    assert np.array_equal(lo, lv[0].cpu().numpy()) and np.array_equal(ro, rv[0].cpu().numpy())
    # the tensors PSMNet receives: /255. in float64, CHW, replicate-padded to multiples of 32, then bf16
    for img_u8, got in ((lo, eng.to_network_input(lv, dtype=torch.bfloat16)), (ro, eng.to_network_input(rv, dtype=torch.bfloat16))):
        x = torch.from_numpy(img_u8 / 255.).permute(2, 0, 1).unsqueeze(0).float()
        ph, pw = (((H // 32) + 1) * 32 - H) % 32, (((W // 32) + 1) * 32 - W) % 32
        ref = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], mode="replicate").to(torch.bfloat16)
        assert got.shape == (1, 3, 544, 960) and torch.equal(got.cpu(), ref)
    # PSMNet's feature maps are 1/4 resolution of the padded input with 32 channels (a stand-in for the CNN output)
    rng = np.random.default_rng(4)
    fl = rng.standard_normal((1, 32, 136, 240)).astype(np.float32)
    fr4 = rng.standard_normal((1, 32, 136, 240)).astype(np.float32)
    ph = 4
    hp = np.pad(fr["hints"], ((ph // 2, ph - ph // 2), (0, 0)), mode="edge")[None, None]     # test.py:199-200 replicate pad
    vp = (hp > 0).astype(np.float32)
    got = eng.psmnet_cost_volume(_dev(eng, fl), _dev(eng, fr4), D, _dev(eng, hp), _dev(eng, vp)).cpu().numpy()
    ref = FO.psmnet_cost_volume(fl, fr4, D, hp, vp)
    assert got.shape == (1, 64, 48, 136, 240)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------
# the reference's own call sequence (models/rsgm/rsgm.py:250-294) over the drop-in natives
# ---------------------------------------------------------------------------------------------------------
def _rsgm_py_sequence(nat, left, left_vpp, right_vpp, dmax, p1=11, p2min=17, alpha=0.5, gamma=35, uniqueness=0.95,
                      subpixel=True):
    """What rsgm.py does around its seven natives, cv2 / numba pieces taken from the (golden-pinned) oracle glue.
    `left` is handed to aggregate_SSE as the padded COLOUR image, exactly as rsgm.py:258,270 does."""
    ht, wt = left.shape[:2]
    pad_ht, pad_wd = (((ht // 16) + 1) * 16 - ht) % 16, (((wt // 16) + 1) * 16 - wt) % 16
    pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
    left, left_vpp, right_vpp = (oracle.pad_reflect(a, pad[2], pad[3], pad[0], pad[1]) for a in (left, left_vpp, right_vpp))
    gl, gr = oracle.rgb2gray(left_vpp), oracle.rgb2gray(right_vpp)                    # _census_transform :8-28
    h, w = gl.shape
    ctl, ctr = np.zeros((h, w), np.uint32), np.zeros((h, w), np.uint32)
    nat.census5x5_SSE(gl, ctl, w, h); nat.census5x5_SSE(gr, ctr, w, h)
    dsi = np.zeros((h, w, dmax), np.uint16)                                           # _hamming_matching :30-46
    nat.costMeasureCensus5x5_xyd_SSE(ctl, ctr, dsi, w, h, dmax, 1)
    agg = np.zeros((h, w, dmax), np.uint16)                                           # _aggregate_dsi :48-63
    nat.aggregate_SSE(left, dsi, agg, w, h, dmax, p1, p2min, alpha, gamma)           # <- H x W x 3 colour image
    d0 = np.zeros((h, w), np.float32)                                                 # _disparity_computation :129-153
    nat.matchWTA_SSE(agg, d0, w, h, dmax, uniqueness)
    nat.subPixelRefine(agg, d0, w, h, dmax, 0)
    fd = np.zeros((h, w), np.float32)
    nat.median3x3_SSE(d0, fd, w, h)
    oracle._linear_interpolate(fd, 15, 3)
    fd = np.clip(fd, 0, None)
    r0 = np.zeros((h, w), np.float32)                                                 # _right_disparity_computation :155-181
    nat.matchWTARight_SSE(agg, r0, w, h, dmax, uniqueness)
    fr = np.zeros((h, w), np.float32)
    nat.median3x3_SSE(r0, fr, w, h)
    oracle._linear_interpolate(fr, 15, 3)
    fr = np.clip(fr, 0, None)
    c = [pad[2], h - pad[3], pad[0], w - pad[1]]
    fd, fr = np.ascontiguousarray(fd[c[0]:c[1], c[2]:c[3]]), np.ascontiguousarray(fr[c[0]:c[1], c[2]:c[3]])
    keep = fd.copy()
    mask = oracle._left_right_check(fd, fr, 1)
    fd[mask == 128] = 0
    f8 = np.ascontiguousarray(fd.astype(np.uint8))
    oracle.filterSpeckles(f8, 0, 200, 10)
    fd = f8.astype(np.float32)
    if subpixel:
        fd[fd != 0] = keep[fd != 0]
    fd = np.ascontiguousarray(fd)
    oracle._interpolate_background(fd)
    return fd


@pytest.mark.parametrize("H,W,D", [(60, 100, 64), (37, 130, 192)])
def test_rsgm_py_call_sequence_with_colour_left_equals_fused_compute_rsgm(H, W, D):
    """INTEGRATION.md's 'keep rsgm.py, swap the natives' route: the stage entry points driven exactly like
    rsgm.py drives pyrSGM (colour `left` into aggregate_SSE) give what the fused compute_rsgm gives."""
    import vppstereo_amd
    from vppstereo_amd import pyrSGM
    fr = synth.make_frame(H, W, D, 0.05, seed=H)
    oracle.init_rand(2)
    lv, rv = oracle.vpp(fr["left"], fr["right"], fr["hints"])
    fused = vppstereo_amd.compute_rsgm(fr["left"], lv, rv, dmax=D)
    staged = _rsgm_py_sequence(pyrSGM, fr["left"], lv, rv, D)
    assert np.array_equal(staged, fused)
    assert np.array_equal(fused, oracle.compute_rsgm(fr["left"], lv, rv, dmax=D))
    # [H,W,1] and [H,W] images are taken as they are; other channel counts are rejected
    g = oracle.rgb2gray(fr["left"])
    g16 = np.ascontiguousarray(np.pad(g, ((0, (16 - H % 16) % 16), (0, (16 - W % 16) % 16)), mode="edge"))
    h, w = g16.shape
    dsi = np.random.default_rng(1).integers(0, 25, (h, w, 64)).astype(np.uint16)
    a, b = np.zeros_like(dsi), np.zeros_like(dsi)
    pyrSGM.aggregate_SSE(g16, dsi, a, w, h, 64, 11, 17, 0.5, 35)
    pyrSGM.aggregate_SSE(g16[..., None], dsi, b, w, h, 64, 11, 17, 0.5, 35)
    assert np.array_equal(a, b)
    with pytest.raises(Exception, match="1 or 3 channels"):
        pyrSGM.aggregate_SSE(np.zeros((h, w, 2), np.uint8), dsi, b, w, h, 64, 11, 17, 0.5, 35)


# ---------------------------------------------------------------------------------------------------------
# robustness
# ---------------------------------------------------------------------------------------------------------
def test_default_stream_ordering_with_torch_work_before_and_after(eng):
    """On torch's default stream (the legacy null stream) the library launches on that stream itself: inputs produced
    by torch kernels just before the call and torch consumers right after it need no synchronisation."""
    import torch
    B, H, W, D = 2, 64, 112, 64
    b = synth.make_batch(B, H, W, D, 0.05, seed=33)
    want = []
    for f in range(B):
        oracle.init_rand(5 + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f])
        want.append(oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D))
    want = np.stack(want)
    assert torch.cuda.current_stream(eng.device).cuda_stream == 0
    big = torch.empty((64, 1024, 1024), dtype=torch.float32, device=eng.device)
    l0, r0, h0 = _dev(eng, b["left"]), _dev(eng, b["right"]), _dev(eng, b["hints"])
    for it in range(5):
        big.normal_()                                            # keeps the stream busy ahead of the producers
        left = (l0.to(torch.int16) + it - it).to(torch.uint8)    # inputs written by torch kernels, no sync
        right = r0.clone()
        hints = h0 * 1.0
        out = eng.vpp_rsgm(left, right, hints, seed=5, rsgm_kw=dict(dmax=D))
        s = out.sum(dtype=torch.float64)                         # torch consumer right behind, no sync
        left.zero_(); right.zero_(); hints.zero_()               # overwrite the inputs behind the library's kernels
        got = out.cpu().numpy()
        assert np.array_equal(got, want), it
        assert abs(float(s) - float(want.astype(np.float64).sum())) < 1e-3
    # and on a side stream
    st = torch.cuda.Stream(device=eng.device)
    with torch.cuda.stream(st):
        left = l0.clone()
        out = eng.vpp_rsgm(left, r0, h0, seed=5, rsgm_kw=dict(dmax=D))
        left.zero_()
        got = out.cpu().numpy()
    assert np.array_equal(got, want)
    torch.cuda.synchronize()


def test_library_calls_leave_the_current_device_alone(eng):
    import torch
    before = torch.cuda.current_device()
    eng.occlusion_heuristic(torch.zeros((1, 8, 8), dtype=torch.float32, device=eng.device))
    assert torch.cuda.current_device() == before


def test_engine_rejects_mismatched_buffers_and_unknown_parameters(eng):
    import torch
    from vppstereo_amd import _lib
    dev = eng.device
    l = torch.zeros((1, 16, 32, 3), dtype=torch.uint8, device=dev)
    h = torch.zeros((1, 16, 32), dtype=torch.float32, device=dev)
    with pytest.raises(ValueError, match="right"):
        eng.vpp_rsgm(l, torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device=dev), h)
    with pytest.raises(ValueError, match="hints"):
        eng.vpp_rsgm(l, l, torch.zeros((1, 16, 31), dtype=torch.float32, device=dev))
    with pytest.raises(ValueError, match="out"):
        eng.vpp_rsgm(l, l, h, out=torch.zeros((1, 16, 32), dtype=torch.float64, device=dev))
    with pytest.raises(ValueError, match="g_occ"):
        eng.vpp_rsgm(l, l, h, g_occ=torch.zeros((1, 16, 32), dtype=torch.float32, device=dev))
    with pytest.raises(ValueError, match="l_vpp"):
        eng.vpp_rsgm(l, l, h, l_vpp=torch.zeros((1, 16, 32, 1), dtype=torch.uint8, device=dev))
    with pytest.raises(ValueError, match="contiguous"):
        eng.vpp_rsgm(l.permute(0, 2, 1, 3), l, h)
    with pytest.raises(TypeError, match="blending"):
        _lib.vpp_params(blending=0.4)          # the reference's keyword is not the C field name (c)
    with pytest.raises(TypeError, match="wsizeAgg_x"):
        eng.vpp(l, l, h, wsizeAgg_x=64)


def test_init_rand_of_either_module_seeds_the_one_stream_vpp_and_the_scans_share():
    """The reference has ONE libc rand() state: init_rand, then scans and vpp() calls in any mix continue it."""
    from vppstereo_amd import vpp_core_opt, vpp_standalone
    fr = synth.make_frame(48, 80, 32, 0.08, seed=3)
    H, W = 48, 80
    occ = np.zeros((H, W), np.uint8)
    # (1) vpp_core_opt.init_rand seeds vpp()
    oracle.init_rand(123)
    lo, ro = oracle.vpp(fr["left"], fr["right"], fr["hints"])
    vpp_core_opt.init_rand(123)
    lg, rg = vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"])
    assert np.array_equal(lo, lg) and np.array_equal(ro, rg)
    # (2) a scan, then vpp(), then a scan: one continuous stream
    oracle.init_rand(9)
    vpp_standalone.init_rand(9)
    a0, b0, a1, b1 = fr["left"].copy(), fr["right"].copy(), fr["left"].copy(), fr["right"].copy()
    for mod, (a, b) in ((oracle, (a0, b0)), (vpp_core_opt, (a1, b1))):
        mod.virtual_projection_scan_rnd(a, b, fr["hints"], W, H, 3, False, 3, 1, 0.4, 0.0, occ, False, True)
    lo, ro = oracle.vpp(fr["left"], fr["right"], fr["hints"], wsize=5)
    lg, rg = vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"], wsize=5)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    assert np.array_equal(lo, lg) and np.array_equal(ro, rg)
    for mod, (a, b) in ((oracle, (a0, b0)), (vpp_core_opt, (a1, b1))):
        mod.virtual_projection_scan_rnd(a, b, fr["hints"], W, H, 3, True, 3, 0, 0.4, 0.0, occ, False, True)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    # (3) non-zero hints that are all <= 0: the reference's `gt[gt>0].min()` raises ValueError
    neg = np.where(fr["hints"] > 0, -1.0, 0.0).astype(np.float32)
    with pytest.raises(ValueError):
        vpp_standalone.vpp(fr["left"], fr["right"], neg)
