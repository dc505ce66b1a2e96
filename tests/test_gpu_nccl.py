"""GPU: RCCL's first contact happens here, not in the driver's multi-GPU run.  A fresh process forms a one-rank `nccl`
process group on the box's GPU and sends the hot path's outputs through the collective branch of
`gather_disparities_async` (force_collective=True) plus the other collectives bench.py uses for N > 1 (test.py:98 is the
reference's only multi-device construct; here frames shard over ranks and the gather to rank 0 is the one exchange)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("B,H,W,D", [(8, 96, 208, 192), (3, 60, 130, 64)])
def test_one_rank_nccl_gather_through_the_collective_branch(B, H, W, D):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_rank.py"), str(_free_port()), str(B), str(H), str(W), str(D)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "nccl one-rank ok" in p.stdout


def test_headline_loop_keeps_its_lock_step_under_rank0s_receive_load():
    """tools/rccl_contention.py: the 32-frame pipelined loop while a one-rank RCCL group moves what rank 0 of an 8-rank run
    receives per step (7 x 66 MB) on the communicator's stream, and while a CU copy kernel moves 464 MB on a side stream.  The
    lock-step kernel shares the chip with those kernels: no launch may lose its lock step, every run must end healthy."""
    import json
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_contention.py"), "", "4"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{\"what\"")][-1])
    assert res["lost_lock_steps_total"] == 0, res
    assert all(r["status"] == "ok" for r in res["runs"]) and len(res["runs"]) == 12
    assert res["slowdown_rccl_7"] < 1.15, res   # (measured 1.04: a regression guard, not a performance claim)
