"""Process of tests/test_gpu_nccl.py (started fresh): ONE rank on the box's one GPU with the `nccl` backend (= RCCL on
ROCm).  Every torch.distributed call bench.py makes for N > 1 ranks is made here with a world of one, the gather through
`force_collective=True`, on tensors Engine.vpp_rsgm produced on torch's legacy default stream, double-buffered like
bench.py's step().  Exit status 0 = everything came back unchanged."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port, B, H, W, D = (int(v) for v in sys.argv[1:6])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from vppstereo_amd import dist as vdist
    from vppstereo_amd.engine import Engine
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # bench.py:383
    try:
        assert dist.get_backend() == "nccl"
        eng = Engine(0)
        eng.set_pipeline(True)
        b = synth.make_batch(B, H, W, D, 0.04, seed=99)
        left, right, hints = (torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("left", "right", "hints"))
        n_out, steps = 2, 5
        outs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(n_out)]
        pending, kept, want_of = [None] * n_out, [], [None] * n_out
        # first contact outside the loop, like bench.py's warm-up gather (RCCL creates its channels on first use)
        vdist.gather_disparities_async(outs[0].zero_(), B, dst=0, force_collective=True).wait()
        torch.cuda.synchronize()
        for s in range(steps):          # bench.py step(): the gather of step k overlaps the kernels of step k+1
            k = s % n_out
            if pending[k] is not None:
                pending[k].wait()
                kept.append((pending[k].result().clone(), want_of[k]))
                pending[k] = None
            eng.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", out=outs[k], seed=7 + s, rsgm_kw=dict(dmax=D, subpixel=1))
            want_of[k] = outs[k].clone()    # same stream, queued behind the call: what the shard holds when the gather reads it
            h = vdist.gather_disparities_async(outs[k], B, dst=0, force_collective=True)
            assert h._work is not None, "the collective branch was not taken"
            pending[k] = h
        for k in range(n_out):
            if pending[k] is not None:
                kept.append((pending[k].result().clone(), want_of[k]))
        eng.synchronize()
        assert eng.uses_vert() == (3 if B >= 3 and D in (128, 192) else 0)
        assert len(kept) == steps
        for got, want in kept:
            assert got.shape == want.shape and torch.equal(got, want) and not bool(torch.isnan(got).any())
        # the other collectives of bench.py's N > 1 path, on device tensors
        flag = torch.tensor([1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                          # bench.py: gather-usable agreement
        t = torch.tensor([3.25], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t)]
        dist.all_gather(allt, t)                                             # bench.py: per-rank step times
        m = torch.tensor([2.5], dtype=torch.float64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        assert int(flag.item()) == 1 and float(allt[0].item()) == 3.25 and float(m.item()) == 2.5
        print("nccl one-rank ok", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
