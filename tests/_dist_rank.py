"""Rank process of tests/test_gpu_dist.py (started fresh, one per rank; the ranks share GPU 0, so the process group is
gloo and the aggregation layout is the line-parallel one: two lock-step launches must not compete for one GPU's block
slots).  Rank r runs the hot path on its contiguous block of the global frames and takes part in the gather to rank 0,
exactly like bench.py's step; rank 0 stores the gathered disparities."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, B, H, W, D, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VPPX_VERT="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from vppstereo_amd import dist as vdist
    from vppstereo_amd.engine import Engine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_total = B * world
        lo, hi = vdist.shard_range(n_total, rank, world)
        eng = Engine(0)
        eng.set_pipeline(True)
        fr = [synth.make_frame(H, W, D, 0.04, seed=77, frame=f) for f in range(lo, hi)]
        left, right, hints = (torch.from_numpy(np.stack([f[k] for f in fr])).to(eng.device) for k in ("left", "right", "hints"))
        outs = []
        for step in range(2):   # two steps, the second gather overlapping nothing but itself: the bench's double buffering
            out = eng.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", seed=vdist.frame_seed(5 + step, lo),
                               rsgm_kw=dict(dmax=D, subpixel=1))
            outs.append(vdist.gather_disparities_async(out, n_total, dst=0))
        eng.synchronize()
        res = [h.result() for h in outs]
        if rank == 0:
            np.save(out_path, torch.stack(res).cpu().numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
