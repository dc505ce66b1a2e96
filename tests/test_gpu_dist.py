"""GPU: sharding independence end to end (test.py:307-311: frames are independent).  Two freshly started rank processes
share the one GPU of the box (gloo process group, line-parallel aggregation), each runs the hot path on its half of the
global frames and the disparities are gathered to rank 0 like in bench.py; the result must equal ONE process running all
frames in one batch (which takes the fused aggregation layout): no dependence on the split, the rank, the layout."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("B,H,W,D", [(4, 96, 200, 192), (3, 60, 130, 64)])
def test_two_ranks_sharing_the_device_equal_one_rank(tmp_path, B, H, W, D):
    import torch
    from vppstereo_amd import dist as vdist
    from vppstereo_amd.engine import Engine
    world, port = 2, _free_port()
    out_path = str(tmp_path / "gathered.npy")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the single-rank run over the same global frames FIRST: it takes the fused lock-step layout, which wants the GPU's
    # block slots to itself (the rank processes would compete for them)
    eng = Engine()
    n_total = B * world
    fr = [synth.make_frame(H, W, D, 0.04, seed=77, frame=f) for f in range(n_total)]
    left, right, hints = (torch.from_numpy(np.stack([f[k] for f in fr])).to(eng.device) for k in ("left", "right", "hints"))
    want = []
    for step in range(2):
        want.append(eng.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", seed=vdist.frame_seed(5 + step, 0),
                                 rsgm_kw=dict(dmax=D, subpixel=1)).clone())
    eng.synchronize()
    assert eng.uses_vert() == (3 if n_total >= 3 and D in (128, 192) else 0)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_rank.py"), str(r), str(world), str(port),
                               str(B), str(H), str(W), str(D), out_path], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out_path)
    assert got.shape == (2, n_total, H, W)
    for step in range(2):
        assert np.array_equal(got[step], want[step].cpu().numpy()), step
