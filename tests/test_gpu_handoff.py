"""GPU parity of the hand-off rows (PSMNet volume + modulation, RAFT correlation
modulation, payload decoders) through the C-ABI, against (1) the golden vectors the reference's own
code produced (tests/golden/frontend_cases.npz) and (2) the numpy oracle on larger seeded inputs.
Tolerance: the only inexact operation is float32 exp(); copies / decoders are bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_cases.npz"))
RTOL, ATOL = 2e-6, 1e-6


@pytest.fixture(scope="module")
def eng():
    import torch
    from vppstereo_amd.engine import Engine
    assert torch.cuda.is_available()
    return Engine()


def _dev(eng, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_psmnet_cost_volume_golden(eng, tag):
    maxdisp, with_hints = (int(v) for v in G[f"psm_{tag}_meta"])
    kw = {}
    if with_hints:
        kw = dict(hints=_dev(eng, G[f"psm_{tag}_hints"]), validhints=_dev(eng, G[f"psm_{tag}_valid"]))
    got = eng.psmnet_cost_volume(_dev(eng, G[f"psm_{tag}_fl"]), _dev(eng, G[f"psm_{tag}_fr"]), maxdisp, **kw).cpu().numpy()
    ref = G[f"psm_{tag}_cost"]
    assert got.shape == ref.shape
    if not with_hints:
        assert np.array_equal(got, ref)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_raft_corr_golden(eng, tag):
    pre = _dev(eng, G[f"raft_{tag}_pre"])
    got = eng.raft_corr_modulate_(pre, _dev(eng, G[f"raft_{tag}_hints"]), _dev(eng, G[f"raft_{tag}_valid"])).cpu().numpy()
    np.testing.assert_allclose(got, G[f"raft_{tag}_post"], rtol=RTOL, atol=ATOL)
    # whole CorrBlock1D.corr (GEMM by the library + modulation): GEMM summation order differs -> looser
    full = eng.raft_corr(_dev(eng, G[f"raft_{tag}_f2"]), _dev(eng, G[f"raft_{tag}_f3"]), _dev(eng, G[f"raft_{tag}_hints"]),
                         _dev(eng, G[f"raft_{tag}_valid"])).cpu().numpy()
    np.testing.assert_allclose(full, G[f"raft_{tag}_post"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 32, 68, 120, 272, 480, 192, 0.05), (1, 8, 33, 61, 135, 247, 64, 0.5),
                                   (1, 5, 7, 10, 30, 43, 16, 1.0)])
def test_psmnet_cost_volume_vs_oracle(eng, shape):
    from oracle import frontends as FO
    B, C, H4, W4, H, W, maxdisp, p = shape
    rng = np.random.default_rng(B * H + W)
    fl = rng.standard_normal((B, C, H4, W4)).astype(np.float32)
    fr = rng.standard_normal((B, C, H4, W4)).astype(np.float32)
    valid = (rng.random((B, 1, H, W)) < p).astype(np.float32)
    hints = rng.uniform(0.5, maxdisp - 1, (B, 1, H, W)).astype(np.float32) * valid
    got = eng.psmnet_cost_volume(_dev(eng, fl), _dev(eng, fr), maxdisp, _dev(eng, hints), _dev(eng, valid)).cpu().numpy()
    ref = FO.psmnet_cost_volume(fl, fr, maxdisp, hints, valid)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)
    # property: hint-free pixels are exact copies of the un-modulated volume
    plain = eng.psmnet_cost_volume(_dev(eng, fl), _dev(eng, fr), maxdisp).cpu().numpy()
    assert np.array_equal(plain, FO.psmnet_cost_volume(fl, fr, maxdisp))
    _, v = FO.subsample_hints(hints, valid)
    keep = np.broadcast_to((v == 0)[:, None, None], got.shape)
    assert np.array_equal(got[keep], plain[keep])


def test_raft_modulate_vs_oracle_and_untouched_rows(eng):
    from oracle import frontends as FO
    B, H4, W2, H, W = 2, 34, 60, 136, 243
    rng = np.random.default_rng(5)
    corr = rng.standard_normal((B, H4, W2, 1, W2)).astype(np.float32)
    valid = (rng.random((B, 1, H, W)) < 0.1).astype(np.float32)
    hints = rng.uniform(0.5, 100, (B, 1, H, W)).astype(np.float32) * valid
    got = eng.raft_corr_modulate_(_dev(eng, corr), _dev(eng, hints), _dev(eng, valid)).cpu().numpy()
    np.testing.assert_allclose(got, FO.raft_corr_modulate(corr, hints, valid), rtol=RTOL, atol=ATOL)
    _, v = FO.subsample_hints(hints, valid)
    assert np.array_equal(got[v == 0], corr[v == 0])


def test_hint_shape_mismatch_is_an_error(eng):
    import torch
    from vppstereo_amd._lib import VppxError
    f = torch.zeros((1, 2, 5, 9), dtype=torch.float32, device=eng.device)
    h = torch.zeros((1, 1, 24, 36), dtype=torch.float32, device=eng.device)   # 24 // 4 = 6 != 5
    with pytest.raises(VppxError):
        eng.psmnet_cost_volume(f, f, 16, h, h)


def test_decoders(eng):
    import torch
    from oracle import frontends as FO
    u16 = G["kitti_u16"]
    d, v = eng.kitti_disp_decode(torch.from_numpy(u16.view(np.int16)).to(eng.device))
    assert np.array_equal(d.cpu().numpy(), G["kitti_disp"]) and np.array_equal(v.cpu().numpy(), G["kitti_valid"])
    rng = np.random.default_rng(3)
    big = rng.integers(0, 65536, (375, 1242), dtype=np.uint16)
    d, v = eng.kitti_disp_decode(torch.from_numpy(big.view(np.int16)).to(eng.device))
    rd, rv = FO.kitti_disp_decode(big)
    assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(v.cpu().numpy(), rv)
    for tag in ("g", "c"):
        H, W, ch, little = (int(x) for x in G[f"pfm_{tag}_meta"])
        got = eng.pfm_decode(_dev(eng, G[f"pfm_{tag}_raw"]), H, W, ch, bool(little)).cpu().numpy()
        assert np.array_equal(got, G[f"pfm_{tag}_dec"])
