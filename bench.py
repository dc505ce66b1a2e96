#!/usr/bin/env python3
"""bench.py -- headline benchmark of the VPP + rSGM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): Mdisparities/s = frames * H * W * D / wall-seconds / 1e6 for
VPP (rnd, reference defaults) + rSGM at 540x960, D=192, 3 % hints; inputs are resident in HBM
when the timed region starts.  A "step" is one pass of the whole hot path (vppx_vpp_rsgm_dev)
over one batch of B synthetic frames per GPU.  Frames shard over ranks with no data-path
collective ("weak" scaling: B frames per GPU); the only exchange is the final gather of the
disparity maps to rank 0, which is inside the timed region for N > 1.

One JSON line is printed by rank 0.  Extra objects:
  roofline     -- dominant kernel (8-path aggregation): algorithmic bytes per launch
                  (10 B/cell of SURVEY 8d's 16 B/cell) / average
                  launch duration measured with hipEvents on the launch stream, vs 8 TB/s.
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm; the reference's own
                  rSGM natives are not in its tree) timed on this box's host cores on a
                  bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, D, P_HINTS, C = 540, 960, 192, 0.03, 3
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
AGG_BYTES_PER_CELL = 10.0        # aggregation share of the 16 B/cell algorithmic bytes
PATH_BYTES_PER_CELL = 16.0       # whole path, + 50 B/pixel
PATH_BYTES_PER_PIXEL = 50.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames timed for the CPU baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--cpu-parallel", action="store_true", help="also time the CPU port on up to 32 host cores")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--shape", type=float, nargs=4, metavar=("H", "W", "D", "P"), default=None,
                    help="other BASELINE configs, e.g. --shape 375 1242 192 0.05 (KITTI) or 1536 2048 256 0.01")
    ap.add_argument("--graph", action="store_true",
                    help="run on a side stream with hipGraph replay of the fused call (small batches are launch-bound)")
    ap.add_argument("--uniform-random", action="store_true",
                    help="uniform-random u8 images and hint values (the variant SURVEY section 6 timed on the CPU) "
                         "instead of the textured scenes; GPU timing only")
    return ap.parse_args()


def _cpu_one_frame(f):
    """VPP (rnd) + rSGM of synthetic frame f with the CPU oracle; returns (seconds, disparity)."""
    import oracle
    import synth
    fr = synth.make_frame(H, W, D, P_HINTS, seed=1234, frame=f)
    t0 = time.perf_counter()
    oracle.init_rand(1 + f)
    lv, rv = oracle.vpp(fr["left"], fr["right"], fr["hints"])
    disp = oracle.compute_rsgm(fr["left"], lv, rv, dmax=D, subpixel=True)
    return time.perf_counter() - t0, disp


def _cpu_worker(f):
    return _cpu_one_frame(f)[0]


def cpu_baseline(n_frames, gpu_out=None):
    """Oracle (CPU restatement of the reference's algorithm), 1 thread, full-size frames -- how the
    reference runs (rsgm.py:44 numThreads=1, Cython scan single-threaded, test.py:293 batch 1)."""
    t_tot, epe = 0.0, None
    for f in range(n_frames):
        dt, disp = _cpu_one_frame(f)
        t_tot += dt
        if gpu_out is not None and f < gpu_out.shape[0]:
            e = float(abs(disp - gpu_out[f]).mean())
            epe = e if epe is None else max(epe, e)
    base = dict(value=n_frames * H * W * D / t_tot / 1e6, unit="Mdisparities/s", cores=1, kind="port",
                sample=f"{n_frames} full {H}x{W}x{D} frames (VPP rnd + rSGM), oracle/liboracle.so gcc -O2, 1 thread, "
                       f"{t_tot / n_frames:.2f} s/frame, host has {os.cpu_count()} cpus")
    return base, epe


def cpu_baseline_parallel(max_procs=32):
    """Same port, frames in parallel over host cores (one process per frame)."""
    import multiprocessing as mp
    n = max(1, min(max_procs, (os.cpu_count() or 1)))
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(n) as pool:
        per = pool.map(_cpu_worker, list(range(n)))
    wall = time.perf_counter() - t0
    return dict(value=n * H * W * D / wall / 1e6, unit="Mdisparities/s", cores=n, kind="port",
                sample=f"{n} frames on {n} processes, wall {wall:.1f} s (includes frame synthesis), "
                       f"mean {sum(per) / n:.2f} s/frame/core")


def main():
    global H, W, D, P_HINTS
    args = parse()
    if os.environ.get("BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_DEBUG"]), exit=True)
    if args.shape:
        H, W, D, P_HINTS = int(args.shape[0]), int(args.shape[1]), int(args.shape[2]), float(args.shape[3])
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from vppstereo_amd import dist as vdist
    from vppstereo_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)  # dry runs: several ranks on one GPU
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", dev_index)
    eng = Engine(dev_index)
    if args.graph:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))  # stream capture needs a non-default stream
        eng.set_graph_mode(True)

    B = args.batch
    n_total = B * world
    lo, hi = vdist.shard_range(n_total, rank, world)
    # synthetic frames: a few distinct scenes tiled over the batch (generation is host-side numpy)
    n_unique = min(B, 4)
    if args.uniform_random:
        trip = [synth.uniform_random_pair(H, W, D, P_HINTS, seed=lo + i) for i in range(n_unique)]
        base = {"left": np.stack([t[0] for t in trip]), "right": np.stack([t[1] for t in trip]),
                "hints": np.stack([t[2] for t in trip])}
        args.cpu_frames = 0  # the CPU leg regenerates the textured frames: not comparable
    else:
        base = synth.make_batch(n_unique, H, W, D, P_HINTS, seed=1234, frame0=lo)
    idx = [i % n_unique for i in range(B)]
    left = torch.from_numpy(np.ascontiguousarray(base["left"][idx])).to(dev)
    right = torch.from_numpy(np.ascontiguousarray(base["right"][idx])).to(dev)
    hints = torch.from_numpy(np.ascontiguousarray(base["hints"][idx])).to(dev)
    outs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
    out = outs[0]
    seed0 = vdist.frame_seed(1, lo)
    pending = [None, None]  # gathers in flight, one per output buffer
    nstep = [0]

    def local_step(buf=None):
        buf = out if buf is None else buf
        eng.vpp_rsgm(left, right, hints, out=buf, seed=seed0, rsgm_kw=dict(dmax=D, subpixel=1))
        return buf

    def step():
        # the gather of step k (to rank 0, asynchronous) overlaps the kernels of step k+1, which write the
        # other output buffer; a buffer is reused only after its previous gather has completed
        k = nstep[0] % 2
        nstep[0] += 1
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        local_step(outs[k])
        if world > 1 and not args.no_gather:
            pending[k] = vdist.gather_disparities_async(outs[k], n_total, dst=0)  # every rank calls step() equally often

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    gather_note = None
    if world > 1 and not args.no_gather:
        # RCCL creates its send/recv channels on first use: do that outside the timed region even with --warmup 0.
        # Should the gather be unusable on this node, every rank agrees to keep its shard local (the data path
        # has no collective; the gather only delivers results to rank 0) and the line says so.
        ok = 1
        try:
            vdist.gather_disparities_async(outs[0], n_total, dst=0).wait()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[bench] rank {rank}: result gather failed ({type(e).__name__}: {e}); running without it", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            args.no_gather = True
            gather_note = "result gather to rank 0 unavailable: shards stay on their ranks"

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.agg_kernel_ms(0)  # reset: the hipEvent pairs around the aggregation launches of the timed steps only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # the last gathers complete inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = n_total * H * W * D * args.steps / dt / 1e6

    result = None
    if rank == 0:
        # ---- dominant kernel, measured live with hipEvents on its own launch stream ----------
        # average over the launches made INSIDE the timed region (event pairs on the launch stream, ring of 64)
        agg_ms, agg_n = eng.agg_kernel_ms(args.steps)
        agg_ms_b2b = eng.time_aggregate(iters=max(3, min(10, args.steps)))  # same kernel re-launched back to back
        agg_frames = eng.time_aggregate_frames()  # frames per launch (the batch is split over sub-streams)
        Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
        cells_launch = agg_frames * Hp * Wp * D
        achieved = cells_launch * AGG_BYTES_PER_CELL / (agg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_traffic.json: separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 FETCH correction
        # applied); only meaningful for the batch size they were collected at
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if pmc.get("batch") == agg_frames and (H, W, D) == (540, 960, 192):
                for name, d in pmc["kernels"].items():
                    if name.startswith(("void sgm_paths_kernel", "sgm_paths_kernel")) and "hbm_GB_per_launch_corrected" in d:
                        traffic = d["hbm_GB_per_launch_corrected"]
        except Exception:
            traffic = None
        eng.enable_stage_timing(True)
        for _ in range(2):  # first pass sizes the un-split workspace; report the second
            local_step()    # rank-0-only section: no collectives here
            torch.cuda.synchronize()
        stages = eng.stage_ms()
        eng.enable_stage_timing(False)
        pipeline_gbs = value * 1e6 * (PATH_BYTES_PER_CELL + PATH_BYTES_PER_PIXEL / D) / 1e9
        result = {
            "metric": f"Mdisparities/s (HxWxD / s) VPP+rSGM at {H}x{W}xD={D}",
            "value": round(value, 1), "unit": "Mdisparities/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u16",
            "data": "synthetic (uniform random u8)" if args.uniform_random else "synthetic",
            "config": {"workload": f"{H}x{W} RGB pair, {100 * P_HINTS:g}% hints, VPP(rnd, wsize 3)+rSGM D={D} subpixel, "
                                   f"{B} frames/GPU/step resident in HBM", "frames_per_step": n_total,
                       "H": H, "W": W, "D": D, "hint_density": P_HINTS},
            "roofline": {"bound": "hbm", "kernel": "sgm_paths_kernel (8-path aggregation)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_unit": "GB per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, committed profile)",
                         "algorithmic_GB_per_launch": round(cells_launch * AGG_BYTES_PER_CELL / 1e9, 3),
                         "kernel_ms": round(agg_ms, 4), "kernel_launches_timed": agg_n,
                         "kernel_ms_back_to_back": round(agg_ms_b2b, 4), "bytes_per_cell": AGG_BYTES_PER_CELL,
                         "frames_per_launch": agg_frames, "cells_per_launch": cells_launch},
            "pipeline_roofline": {"bytes_per_cell": PATH_BYTES_PER_CELL, "bytes_per_pixel": PATH_BYTES_PER_PIXEL,
                                  "achieved_GBps": round(pipeline_gbs, 1),
                                  "frac": round(pipeline_gbs / HBM_PEAK_GBS, 4)},
            "stage_ms": {k: round(v, 3) for k, v in stages.items()},
            "device": eng.ctx.device_name,
        }
        if args.graph:
            result["graph_replays"] = eng.graph_replays()
        if world > 1:
            result["config"]["result_gather"] = gather_note or ("off" if args.no_gather else "async gather to rank 0, overlapped")
        if world == 1 and args.cpu_frames > 0:
            gpu_out = out[: min(args.cpu_frames, n_unique)].cpu().numpy()
            result["cpu_baseline"], epe = cpu_baseline(min(args.cpu_frames, n_unique), gpu_out)
            result["epe_vs_cpu_oracle"] = epe      # mean |disp_gpu - disp_cpu|, worst frame (0.0 = bit-equal)
            result["speedup_vs_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
            if args.cpu_parallel:
                result["cpu_baseline_all_cores"] = cpu_baseline_parallel()
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
