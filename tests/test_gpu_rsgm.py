"""GPU parity, rSGM: every HIP stage (through the pyrSGM-compatible C-ABI entry points) and
the fused compute_rsgm against the CPU oracle on the same seeded inputs.  Integer stages are
bit-exact; float disparities must agree within 1e-4 (north_star) -- in practice bit-exact."""
import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4  # float disparity tolerance stated by BASELINE.json north_star


@pytest.fixture(scope="module")
def nat():
    from vppstereo_amd import pyrSGM
    return pyrSGM


def _scene(H, W, D, seed=3):
    fr = synth.make_frame(H, W, D, 0.05, seed=seed)
    gl = oracle.rgb2gray(fr["left"])
    gr = oracle.rgb2gray(fr["right"])
    return fr, gl, gr


def _stages(mod, gl, gr, W, H, D, p=(11, 17, 0.5, 35), uniq=0.95):
    cl = np.zeros((H, W), np.uint32); cr = np.zeros((H, W), np.uint32)
    mod.census5x5_SSE(gl, cl, W, H); mod.census5x5_SSE(gr, cr, W, H)
    dsi = np.zeros((H, W, D), np.uint16)
    mod.costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, W, H, D, 1)
    S = np.zeros((H, W, D), np.uint16)
    mod.aggregate_SSE(gl, dsi, S, W, H, D, p[0], p[1], p[2], p[3])
    dl = np.zeros((H, W), np.float32)
    mod.matchWTA_SSE(S, dl, W, H, D, uniq)
    dls = dl.copy()
    mod.subPixelRefine(S, dls, W, H, D, 0)
    dm = np.zeros((H, W), np.float32)
    mod.median3x3_SSE(dls, dm, W, H)
    dr = np.zeros((H, W), np.float32)
    mod.matchWTARight_SSE(S, dr, W, H, D, uniq)
    return dict(cl=cl, cr=cr, dsi=dsi, S=S, dl=dl, dls=dls, dm=dm, dr=dr)


@pytest.mark.parametrize("H,W,D", [(48, 96, 64), (32, 208, 192), (40, 64, 24), (23, 48, 128), (16, 272, 256), (48, 80, 8)])
def test_each_stage_vs_oracle(nat, H, W, D):
    fr, gl, gr = _scene(H, W, D)
    o = _stages(oracle, gl, gr, W, H, D)
    g = _stages(nat, gl, gr, W, H, D)
    for k in ("cl", "cr", "dsi", "S"):
        assert np.array_equal(o[k], g[k]), (k, int((o[k] != g[k]).sum()))
    for k in ("dl", "dr"):
        assert np.array_equal(o[k], g[k]), (k, int((o[k] != g[k]).sum()))
    for k in ("dls", "dm"):
        assert np.max(np.abs(o[k] - g[k])) <= TOL, k
        assert np.array_equal(o[k], g[k]), k  # stricter: identical IEEE operations on both sides
    if W >= D:  # a degenerate frame (narrower than the search range) has no confident matches
        assert (o["dl"] == -10).any() and (o["dl"] > 0).any()


def test_aggregate_single_paths_and_saturation(nat):
    """Large P1/P2 drive the u16 saturating arithmetic; compare against the oracle."""
    H, W, D = 32, 64, 64
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    dsi = rng.integers(0, 60000, (H, W, D)).astype(np.uint16)
    for (p1, p2min, alpha, gamma) in [(11, 17, 0.5, 35), (40000, 50000, 0.5, 60000), (0, 0, 2.0, 5), (300, 20, 0.25, 900)]:
        So = np.zeros((H, W, D), np.uint16); Sg = np.zeros((H, W, D), np.uint16)
        oracle.aggregate_SSE(img, dsi, So, W, H, D, p1, p2min, alpha, gamma)
        nat.aggregate_SSE(img, dsi, Sg, W, H, D, p1, p2min, alpha, gamma)
        assert np.array_equal(So, Sg), (p1, p2min, alpha, gamma, int((So != Sg).sum()))


def test_reference_exceptions(nat):
    import vppstereo_amd
    z = np.zeros((16, 20), np.uint8)
    with pytest.raises(Exception, match=r"Invalid width \(20\): width % 16 != 0"):
        nat.census5x5_SSE(z, np.zeros((16, 20), np.uint32), 20, 16)
    im = np.zeros((16, 32, 3), np.uint8)
    with pytest.raises(Exception, match=r"Invalid dmax \(60\): dmax % 8 != 0"):
        vppstereo_amd.compute_rsgm(im, im, im, dmax=60)
    with pytest.raises(Exception, match=r"Invalid dmax \(264\): dmax > 256"):
        vppstereo_amd.compute_rsgm(im, im, im, dmax=264)
    with pytest.raises(Exception, match=r"Invalid uniqueness"):
        vppstereo_amd.compute_rsgm(im, im, im, dmax=64, uniqueness=1.5)


@pytest.mark.parametrize("H,W,D,C,sub", [(60, 100, 64, 3, True), (60, 100, 64, 3, False), (37, 130, 192, 3, True),
                                         (48, 96, 32, 1, True), (75, 91, 128, 3, False)])
def test_compute_rsgm_vs_oracle(H, W, D, C, sub):
    import vppstereo_amd
    fr = synth.make_frame(H, W, D, 0.05, seed=H + W, channels=C)
    oracle.init_rand(1)
    lv, rv = oracle.vpp(fr["left"], fr["right"], fr["hints"])
    if C == 1:
        l, lv, rv = fr["left"][..., 0], lv[..., 0], rv[..., 0]
    else:
        l = fr["left"]
    want = oracle.compute_rsgm(l, lv, rv, dmax=D, subpixel=sub)
    got = vppstereo_amd.compute_rsgm(l, lv, rv, dmax=D, subpixel=sub)
    assert got.shape == (H, W) and got.dtype == np.float32
    assert np.max(np.abs(want - got)) <= TOL, float(np.max(np.abs(want - got)))
    assert np.array_equal(want, got)
    if not sub:
        assert np.array_equal(got, np.round(got))  # SURVEY C-13: integer-valued without sub-pixel
    # the synthetic scene must be matched sensibly (guards against a degenerate pipeline)
    if W >= 1.5 * D:
        err = np.abs(got - fr["gt"])
        assert np.median(err) < 2.0


def test_compute_rsgm_guided_vs_oracle():
    """--guided: _guided_dsi (rsgm.py:116-127) re-weights the cost rows of the hint pixels."""
    import vppstereo_amd
    H, W, D = 50, 116, 64
    fr = synth.make_frame(H, W, D, 0.06, seed=12)
    valid = (fr["hints"] > 0).astype(np.float32)
    want = oracle.compute_rsgm(fr["left"], fr["left"], fr["right"], hints=fr["hints"], validhints=valid, dmax=D)
    got = vppstereo_amd.compute_rsgm(fr["left"], fr["left"], fr["right"], hints=fr["hints"], validhints=valid, dmax=D)
    plain = vppstereo_amd.compute_rsgm(fr["left"], fr["left"], fr["right"], dmax=D)
    assert np.max(np.abs(want - got)) <= TOL and np.array_equal(want, got)
    assert not np.array_equal(got, plain)


def test_fused_batched_hot_path_vs_oracle():
    """vppx_vpp_rsgm_dev (device-resident, batched) == per-frame oracle vpp -> compute_rsgm."""
    import torch
    from vppstereo_amd.engine import Engine
    B, H, W, D = 3, 52, 120, 64
    b = synth.make_batch(B, H, W, D, 0.04, seed=21)
    eng = Engine()
    dev = eng.device
    hints = torch.from_numpy(b["hints"]).to(dev)
    occ = eng.occlusion_heuristic(hints)
    lv = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    rv = torch.empty_like(lv)
    out = eng.vpp_rsgm(torch.from_numpy(b["left"]).to(dev), torch.from_numpy(b["right"]).to(dev), hints, g_occ=occ,
                       l_vpp=lv, r_vpp=rv, seed=500, vpp_kw=dict(c_occ=0.1), rsgm_kw=dict(dmax=D, subpixel=1))
    torch.cuda.synchronize()
    out, lv, rv, occ = out.cpu().numpy(), lv.cpu().numpy(), rv.cpu().numpy(), occ.cpu().numpy()
    for f in range(B):
        _, conf = oracle.occlusion_heuristic(b["hints"][f])
        assert np.array_equal(conf, occ[f])
        oracle.init_rand(500 + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f], g_occ=conf, c_occ=0.1)
        assert np.array_equal(lo, lv[f]) and np.array_equal(ro, rv[f])
        want = oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D, subpixel=True)
        assert np.max(np.abs(want - out[f])) <= TOL
        assert np.array_equal(want, out[f])


def test_glue_golden_vectors_through_gpu_pipeline():
    """The rsgm.py glue pinned by golden vectors (linear interpolate / LR check / background)
    is exercised inside compute_rsgm; here the speckle filter and background fill are checked
    on a crafted disparity field via the oracle == GPU equality of the full pipeline on a scene
    with large invalid regions."""
    import vppstereo_amd
    rng = np.random.default_rng(0)
    H, W, D = 64, 128, 64
    l = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)   # pure noise: mostly invalid matches, many speckles
    r = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    r[:, :80] = l[:, 10:90]                                # a consistent region with d = 10
    want = oracle.compute_rsgm(l, l, r, dmax=D, subpixel=True)
    got = vppstereo_amd.compute_rsgm(l, l, r, dmax=D, subpixel=True)
    assert np.array_equal(want, got)


def test_network_input_handoff_matches_reference_glue():
    """test.py:179-200: u8 HWC -> /255. -> CHW float -> replicate pad to /32 (and a bf16 variant)."""
    import torch
    import torch.nn.functional as F
    from vppstereo_amd.engine import Engine
    eng = Engine()
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (2, 37, 50, 3), dtype=np.uint8)
    a[0, :4, :4] = np.arange(48, dtype=np.uint8).reshape(4, 4, 3)
    t = torch.from_numpy(a).to(eng.device)
    got = eng.to_network_input(t).cpu()
    refs = []
    for f in range(2):
        x = torch.from_numpy(a[f] / 255.).permute(2, 0, 1).unsqueeze(0).float()          # test.py:179
        ht, wt = x.shape[-2:]
        pad_ht = (((ht // 32) + 1) * 32 - ht) % 32
        pad_wd = (((wt // 32) + 1) * 32 - wt) % 32
        pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]       # test.py:189-196
        refs.append(F.pad(x, pad, mode='replicate'))
    ref = torch.cat(refs, 0)
    assert got.shape == ref.shape and torch.equal(got, ref)
    gb = eng.to_network_input(t, dtype=torch.bfloat16).cpu()
    assert torch.equal(gb, ref.to(torch.bfloat16))
    # all 256 levels survive the float round trip
    lv = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16, 1).to(eng.device)
    back = (eng.to_network_input(lv, pad_multiple=1).cpu() * 255).numpy().astype(np.uint8).reshape(-1)
    assert np.array_equal(back, np.arange(256, dtype=np.uint8))


def test_stage_api_between_fused_calls_does_not_poison_the_penalty_table():
    """The fused path caches its P2 table on the device; aggregate_SSE (stage API) writes another one into
    the same buffer: the next fused call must re-upload its own."""
    from vppstereo_amd import pyrSGM
    from vppstereo_amd.rsgm import compute_rsgm
    fr = synth.make_frame(40, 64, 32, 0.05, seed=8)
    oracle.init_rand(3)
    lv, rv = oracle.vpp(fr["left"], fr["right"], fr["hints"])
    want = oracle.compute_rsgm(fr["left"], lv, rv, dmax=32)
    assert np.array_equal(compute_rsgm(fr["left"], lv, rv, dmax=32), want)
    gl = oracle.rgb2gray(fr["left"])
    gl16 = np.ascontiguousarray(np.pad(gl, ((4, 4), (0, 0)), mode="edge"))        # 48 x 64
    dsi = np.random.default_rng(1).integers(0, 25, (48, 64, 32)).astype(np.uint16)
    S = np.zeros_like(dsi)
    pyrSGM.aggregate_SSE(gl16, dsi, S, 64, 48, 32, 3, 90, 2.0, 200)               # very different penalties
    assert np.array_equal(compute_rsgm(fr["left"], lv, rv, dmax=32), want)
