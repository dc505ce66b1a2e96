"""oracle/frontends.py (numpy restatement of the hand-off rows) against golden vectors
produced by the reference's own code (tests/golden/make_frontend_golden.py)."""
import os

import numpy as np
import pytest

from oracle import frontends as FO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_cases.npz"))
RTOL, ATOL = 2e-6, 1e-6   # float32 exp of two different libms (torch/sleef vs numpy)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_psmnet_cost_volume_matches_reference(tag):
    maxdisp, with_hints = (int(v) for v in G[f"psm_{tag}_meta"])
    kw = dict(hints=G[f"psm_{tag}_hints"], validhints=G[f"psm_{tag}_valid"]) if with_hints else {}
    got = FO.psmnet_cost_volume(G[f"psm_{tag}_fl"], G[f"psm_{tag}_fr"], maxdisp, **kw)
    ref = G[f"psm_{tag}_cost"]
    assert got.shape == ref.shape and got.dtype == np.float32
    if not with_hints:
        assert np.array_equal(got, ref)          # pure copies
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_raft_corr_modulation_matches_reference(tag):
    got = FO.raft_corr_modulate(G[f"raft_{tag}_pre"], G[f"raft_{tag}_hints"], G[f"raft_{tag}_valid"])
    np.testing.assert_allclose(got, G[f"raft_{tag}_post"], rtol=RTOL, atol=ATOL)


def test_nearest_indices_match_torch_upsample():
    import torch
    import torch.nn.functional as F
    for n_in in (20, 22, 23, 37, 375, 540, 1242):
        x = torch.arange(n_in, dtype=torch.float32).view(1, 1, 1, n_in)
        ref = F.interpolate(x, size=[1, n_in // 4], mode="nearest").view(-1).numpy().astype(np.int64)
        assert np.array_equal(FO.nearest_indices(n_in, n_in // 4), ref), n_in


def test_decoders_match_reference():
    d, v = FO.kitti_disp_decode(G["kitti_u16"])
    assert np.array_equal(d, G["kitti_disp"]) and np.array_equal(v, G["kitti_valid"])
    for tag in ("g", "c"):
        H, W, ch, little = (int(x) for x in G[f"pfm_{tag}_meta"])
        got = FO.pfm_decode(G[f"pfm_{tag}_raw"].tobytes(), H, W, ch, bool(little))
        assert np.array_equal(got, G[f"pfm_{tag}_dec"])


def test_losses_match_reference():
    """vppstereo_amd.losses (rows a20) against losses.py:5-24 run by the golden script."""
    import torch
    from vppstereo_amd import losses
    torch.manual_seed(123)
    nh, nv = losses.sample_hints(torch.from_numpy(G["loss_hints"].copy()), torch.from_numpy(G["loss_valid"].copy()), 0.4)
    assert np.array_equal(nv.numpy(), G["loss_new_valid"])
    assert np.array_equal(nh.numpy(), G["loss_new_hints"], equal_nan=True)
    m = losses.guided_metrics(G["met_disp"].copy(), G["met_gt"].copy(), G["met_valid"].copy())
    for k, v in m.items():
        ref = G["met_" + k.replace(" ", "_").replace(".", "p")]
        assert np.array_equal(np.asarray(v), ref), k
