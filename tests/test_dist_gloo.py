"""world_size-2 gloo test of the only multi-GPU step of the path: the final gather of the
per-rank disparity shards (vppstereo_amd/dist.py).  Runs on CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from vppstereo_amd import dist as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        lo, hi = vdist.shard_range(n_frames, rank, ws)
        # frame f's "disparity" is a function of the GLOBAL frame index and its own seed
        local = torch.stack([torch.full((3, 5), float(vdist.frame_seed(100, f))) for f in range(lo, hi)]) \
            if hi > lo else torch.zeros((0, 3, 5))
        full = vdist.gather_disparities(local, n_frames, dst=0)
        # two overlapping asynchronous gathers (what bench.py keeps in flight), completed out of order
        h1 = vdist.gather_disparities_async(local + 1.0, n_frames, dst=0)
        h2 = vdist.gather_disparities_async(local + 2.0, n_frames, dst=0)
        r2, r1 = h2.result(), h1.result()
        if rank == 0:
            assert torch.equal(r1, full + 1.0) and torch.equal(r2, full + 2.0)
            q.put(full.numpy())
        else:
            assert full is None and r1 is None and r2 is None
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8])
def test_gather_world_size_2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = np.stack([np.full((3, 5), float(100 + f), np.float32) for f in range(n_frames)])
    assert np.array_equal(out, want)


def test_gather_identity_without_process_group():
    t = torch.arange(12.0).reshape(2, 2, 3)
    assert vdist.gather_disparities(t, 2) is t
