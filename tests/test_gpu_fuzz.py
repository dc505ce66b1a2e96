"""Seeded differential test: random shapes and parameters of vpp() and compute_rsgm() (the two
drop-in entry points) through the C-ABI against the CPU oracle, bit-exact.  Deterministic seeds;
complements the fixed cases with combinations nobody wrote down."""
import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(6, 70)), int(rng.integers(8, 200))
    C3 = int(rng.choice([1, 3]))
    D = int(rng.choice([8, 16, 24, 40, 64, 96, 128, 192, 256]))
    dens = float(rng.choice([0.0, 0.01, 0.05, 0.2, 0.6]))
    fr = synth.make_frame(H, W, max(D, 8), dens, seed=seed, channels=C3)
    return rng, H, W, C3, D, fr


@pytest.mark.parametrize("seed", list(range(100, 160)))
def test_vpp_random_parameters(seed):
    from vppstereo_amd import vpp_standalone
    rng, H, W, C3, D, fr = _case(seed)
    occ = ((rng.random((H, W)) < 0.3) & (fr["hints"] > 0)).astype(np.uint8)
    kw = dict(wsize=int(rng.choice([1, 3, 5, 7, 9])), left2right=bool(rng.integers(2)), blending=float(rng.choice([0.4, 0.0, 1.0, 0.73])),
              uniform_color=bool(rng.integers(2)), c_occ=float(rng.choice([0.0, 0.2, 1.0])), discard_occ=bool(rng.integers(2)),
              interpolate=bool(rng.integers(2)), g_occ=occ if rng.integers(2) else None,
              use_distance_patch=bool(rng.integers(2)), use_bilateral_patch=bool(rng.integers(4) == 0))
    if rng.integers(5) == 0:
        kw.update(method="maxDistance", wsizeAgg_x=int(rng.choice([5, 17, 64])), wsizeAgg_y=int(rng.choice([1, 3])))
        kw["wsize"] = min(kw["wsize"], 5)
    pos = fr["hints"][fr["hints"] > 0]
    if kw["use_distance_patch"] and pos.size and not pos.max() > pos.min():
        # one distinct hint value: the reference divides by dmax - dmin == 0 under numba's python error model
        oracle.init_rand(seed)
        with pytest.raises(ZeroDivisionError):
            oracle.vpp(fr["left"], fr["right"], fr["hints"], **kw)
        with pytest.raises(ZeroDivisionError):
            vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"], **kw)
        return
    l, r = (fr["left"][..., 0], fr["right"][..., 0]) if (C3 == 1 and rng.integers(2)) else (fr["left"], fr["right"])
    oracle.init_rand(seed)
    lo, ro = oracle.vpp(l, r, fr["hints"], **kw)
    vpp_standalone.init_rand(seed)
    lg, rg = vpp_standalone.vpp(l, r, fr["hints"], **kw)
    assert lg.shape == lo.shape and np.array_equal(lo, lg) and np.array_equal(ro, rg), (seed, {k: v for k, v in kw.items() if k != "g_occ"})


@pytest.mark.parametrize("seed", list(range(200, 260)))
def test_compute_rsgm_random_parameters(seed):
    from vppstereo_amd.rsgm import compute_rsgm
    rng, H, W, C3, D, fr = _case(seed)
    oracle.init_rand(seed)
    lv, rv = oracle.vpp(fr["left"], fr["right"], fr["hints"])
    kw = dict(dmax=D, p1=int(rng.choice([0, 3, 11, 40])), p2min=int(rng.choice([0, 17, 60, 250])), alpha=float(rng.choice([0.0, 0.5, 2.0])),
              gamma=int(rng.choice([0, 35, 100, 300])), uniqueness=float(rng.choice([0.95, 1.0, 0.5, 0.8])), subpixel=bool(rng.integers(2)))
    guided = rng.integers(4) == 0
    if guided:
        kw.update(hints=fr["hints"], validhints=(fr["hints"] > 0).astype(np.float32))
    if C3 == 1:
        args = (fr["left"][..., 0], lv[..., 0], rv[..., 0])
    else:
        args = (fr["left"], lv, rv)
    want = oracle.compute_rsgm(*args, **kw)
    got = compute_rsgm(*args, **kw)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(want, got), (seed, kw.get("dmax"), float(np.max(np.abs(want - got))))


@pytest.mark.parametrize("seed", list(range(300, 316)))
def test_fused_batched_random_parameters(seed):
    """Engine.vpp_rsgm (the call bench.py times) with random batch sizes (both lane layouts of the
    aggregation: B < 8 and B >= 8, XCD-aware block map at B % 8 == 0), shapes and parameters."""
    import torch
    from vppstereo_amd.engine import Engine
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 2, 5, 8, 9, 16]))
    H, W = int(rng.integers(8, 48)), int(rng.integers(16, 120))
    D = int(rng.choice([16, 64, 128, 192, 256]))
    dens = float(rng.choice([0.02, 0.1, 0.5]))
    b = synth.make_batch(B, H, W, max(D, 8), dens, seed=seed)
    occ = ((rng.random((B, H, W)) < 0.25) & (b["hints"] > 0)).astype(np.uint8)
    vkw = dict(wsize=int(rng.choice([1, 3, 5])), direction=int(rng.integers(2)), uniform_color=int(rng.integers(2)),
               interpolate=int(rng.integers(2)), discard_occluded=int(rng.integers(2)), c_occ=float(rng.choice([0.0, 0.3])))
    rkw = dict(dmax=D, p1=int(rng.choice([5, 11])), p2min=int(rng.choice([17, 40])), gamma=int(rng.choice([35, 80])),
               uniqueness=float(rng.choice([0.95, 0.7])), subpixel=int(rng.integers(2)))
    eng = Engine()
    dev = eng.device
    lv = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    rv = torch.empty_like(lv)
    out = eng.vpp_rsgm(torch.from_numpy(b["left"]).to(dev), torch.from_numpy(b["right"]).to(dev), torch.from_numpy(b["hints"]).to(dev),
                       g_occ=torch.from_numpy(occ).to(dev), l_vpp=lv, r_vpp=rv, seed=seed, vpp_kw=vkw, rsgm_kw=rkw)
    torch.cuda.synchronize()
    out, lv, rv = out.cpu().numpy(), lv.cpu().numpy(), rv.cpu().numpy()
    for f in range(B):
        oracle.init_rand(seed + f)
        a0, b0 = b["left"][f].copy(), b["right"][f].copy()
        oracle.virtual_projection_scan_rnd(a0, b0, b["hints"][f], W, H, 3, bool(vkw["uniform_color"]), vkw["wsize"], vkw["direction"],
                                           0.4, vkw["c_occ"], occ[f], bool(vkw["discard_occluded"]), bool(vkw["interpolate"]))
        assert np.array_equal(a0, lv[f]) and np.array_equal(b0, rv[f]), (seed, f, vkw)
        want = oracle.compute_rsgm(b["left"][f], a0, b0, dmax=D, p1=rkw["p1"], p2min=rkw["p2min"], gamma=rkw["gamma"],
                                   uniqueness=rkw["uniqueness"], subpixel=bool(rkw["subpixel"]))
        assert np.array_equal(want, out[f]), (seed, f, B, H, W, D, rkw)
