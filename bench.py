#!/usr/bin/env python3
"""bench.py -- headline benchmark of the VPP + rSGM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): Mdisparities/s = frames * H * W * D / wall-seconds / 1e6 for
occlusion heuristic + VPP (rnd, reference defaults) + rSGM at 540x960, D=192, 3 % hints; inputs are
resident in HBM when the timed region starts.  A "step" is one pass of the whole hot path
(filter.occlusion_heuristic -> vpp -> compute_rsgm, i.e. test.py:154-225 with --maskocc) over one batch of B
synthetic frames per GPU.  Frames shard over ranks with no data-path collective ("weak" scaling: B frames
per GPU); the only exchange is the final gather of the disparity maps to rank 0, inside the timed region.

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts N rank processes itself, before
this process has made any GPU call, and exits with the first non-zero rank status.

One JSON line is printed by rank 0.  Extra objects:
  roofline     -- dominant kernel (the fused vertical aggregation kernel, 6 of the 8 paths; the 8-path kernel
                  when that layout runs): algorithmic bytes per launch (SURVEY 8d's 10 B/cell for the 8-path
                  aggregation, pro rata for the paths the launch carries) / average launch
                  duration measured with hipEvents on the launch stream, vs 8 TB/s; next to it the PMC view of
                  the same kernel (real HBM bytes, VALU issue time) from the committed profile of THIS kernel
                  source (else null).
  b1           -- the literal cfg-2 "single pair": latency of one frame per call.
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm; the reference's own rSGM natives
                  are not in its tree) timed on this box's host cores on a bounded sample, rank 0, N=1.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
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
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06_pmc.json")
N_SIMD = 1024                    # 256 CUs x 4 SIMDs
# average issue cost of one VALU wave-instruction of the FUSED VERTICAL kernel's row loop (and of no other kernel: it is only
# applied to the dominant kernel's own instruction count): its mix of full-rate (~2.5 cycles) and half-rate (~4 cycles:
# VOP3P, three-operand VOP3, v_bcnt, DPP) instructions priced with the rates tools/valu_bench.hip measured on gfx950
# (profiles/r02_valu_issue_rates.txt)
VALU_CYCLES_PER_INST = 3.3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames timed for the 1-thread CPU baseline (0 = skip both CPU legs)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-cpu-parallel", action="store_true", help="skip the all-cores CPU leg")
    ap.add_argument("--no-occ", action="store_true", help="headline run without the occlusion mask (g_occ = None)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend for N > 1 (default nccl = RCCL; gloo when the "
                                                    "ranks have to share a GPU: dry runs)")
    ap.add_argument("--shape", type=float, nargs=4, metavar=("H", "W", "D", "P"), default=None,
                    help="other BASELINE configs, e.g. --shape 375 1242 192 0.05 (KITTI) or 1536 2048 256 0.01")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap the front stage of step k+1 (occlusion heuristic, VPP, pad, census) with the "
                         "sum / WTA and post kernels of step k (Engine.set_pipeline)")
    ap.add_argument("--graph", action="store_true",
                    help="run on a side stream with hipGraph replay of the fused call (small batches are launch-bound)")
    ap.add_argument("--uniform-random", action="store_true",
                    help="uniform-random u8 images and hint values (the variant SURVEY section 6 timed on the CPU) "
                         "instead of the textured scenes; GPU timing only")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the bounded runs of the other BASELINE configurations (cfg 3 KITTI B=32, cfg 5 1536x2048x256 B=8)")
    ap.add_argument("--sustained-steps", type=int, default=500,
                    help="further steps timed AFTER the K timed ones (power steady state of the chip; 0 = skip); reported as "
                         "ms_per_step_sustained, never as value")
    ap.add_argument("--spawn-check", action="store_true",
                    help="rank processes only form the process group (gloo, CPU) and report the frame plan: launcher self-test")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# N-rank launch and the frame plan (no GPU call in here)
# ------------------------------------------------------------------------------------------------
def frame_plan(batch, rank, world, base_seed=1):
    """Frames of one step: `batch` per rank (weak scaling), rank r owns the contiguous global frames [lo, hi);
    frame f draws from srand(base_seed + f) wherever it runs."""
    from vppstereo_amd import dist as vdist
    n_total = batch * world
    lo, hi = vdist.shard_range(n_total, rank, world)
    return dict(n_total=n_total, lo=lo, hi=hi, seed0=vdist.frame_seed(base_seed, lo))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start n fresh rank processes of this script (one per GPU) and wait for them.  Nothing in this process has
    touched the GPU: `torch.cuda.device_count()` does not initialise it on this image."""
    extra = []
    if "--backend" not in argv and "--spawn-check" not in argv:
        try:
            import torch
            if torch.cuda.device_count() < n:
                extra = ["--backend", "gloo"]  # ranks must share a device: RCCL refuses duplicate GPUs
                # ... and the fused vertical kernel wants a GPU's block slots to itself: ranks that
                # share a device (dry runs only) use the 8-path layout
                os.environ.setdefault("VPPX_VERT", "0")
        except Exception:
            pass
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv) + extra, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def pin_to_gpu_numa(torch, dev_index):
    """One process per GPU: keep the rank's host threads on the NUMA node its GPU hangs off (PCIe root complex), so that the
    launch thread and RCCL's proxy thread do not wander over the sockets of a 2-socket host.  Returns what was done."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return {"pci": bdf, "numa_node": node, "pinned": False}
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "pinned": bool(cpus), "cpus": len(cpus)}
    except Exception as e:  # noqa: BLE001
        return {"pinned": False, "why": f"{type(e).__name__}: {e}"[:80]}


# What the first multi-GPU run is expected to show (written down BEFORE it: no multi-GPU node was available to any round).
# Inputs: this run's own one-GPU step time; the one-GPU emulation of rank 0's receive load (profiles/r05_rccl_contention.json:
# 7 inbound shards cost the receiving rank's loop 4.0 %, 1 shard 0.4 % -- taken as linear in the number of senders); the
# gather's size (one float32 disparity shard per sender and step) over one xGMI link each (~153 GB/s per link and direction
# pair, /opt/skills/guides/MI355X_MICROARCH.md), asynchronous and double-buffered, i.e. exposed only where it outlasts a step.
XGMI_LINK_GBS = 153.0
ROOT_SLOWDOWN_PER_SENDER = 0.04 / 7
NEAR_LINEAR_EFFICIENCY = 0.93


def predict_scaling(ms_per_step_1gpu, batch, h, w, d, worlds=(1, 2, 4, 8)):
    """Predicted weak-scaling curve of `bench.py --gpus N`: per N the step time of the slowest rank (rank 0: it receives), the
    gather's bytes and duration, the part of it a step cannot hide, whole-job throughput and efficiency against N x one GPU."""
    shard_bytes = batch * h * w * 4
    rows = []
    for n in worlds:
        root_ms = ms_per_step_1gpu * (1.0 + ROOT_SLOWDOWN_PER_SENDER * (n - 1))
        gather_ms = shard_bytes / (XGMI_LINK_GBS * 1e9) * 1e3 if n > 1 else 0.0   # one link per sender, all in parallel
        exposed = max(0.0, gather_ms - root_ms)
        step_ms = root_ms + exposed
        eff = ms_per_step_1gpu / step_ms
        rows.append({"n_gpus": n, "ms_per_step_rank0": round(root_ms, 3), "ms_per_step_other_ranks": round(ms_per_step_1gpu, 3) if n > 1 else None,
                     "gather_bytes_into_rank0": shard_bytes * (n - 1), "gather_ms_per_link": round(gather_ms, 3),
                     "gather_exposed_ms": round(exposed, 3), "ms_per_step": round(step_ms, 3),
                     "value_Mdisp_per_s": round(n * batch * h * w * d / step_ms / 1e3, 1), "efficiency": round(eff, 4)})
    return {"model": "rank 0's loop slows by 4 % / 7 per sender (one-GPU emulation, profiles/r05_rccl_contention.json); the gather (one "
                     "float32 shard per sender over its own xGMI link at ~153 GB/s) runs under the next step and is exposed only "
                     "beyond a step; the other ranks run at the one-GPU rate; step = slowest rank",
            "near_linear_means_efficiency_at_8_of_at_least": NEAR_LINEAR_EFFICIENCY, "from_ms_per_step_1gpu": round(ms_per_step_1gpu, 3),
            "per_world": rows}


def spawn_check(args):
    """Launcher self-test (CPU, gloo): every rank reports its frame plan and what pinning to its GPU's NUMA node did (nothing,
    without a GPU -- but the path runs), rank 0 prints what it saw."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    plan = frame_plan(args.batch, rank, world)
    pin = pin_to_gpu_numa(torch, int(os.environ.get("LOCAL_RANK", "0")))
    seen = 1
    rows = [[plan["lo"], plan["hi"], plan["seed0"], int(bool(pin.get("pinned")))]]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = dist.get_world_size()
        t = torch.tensor(rows[0], dtype=torch.int64)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        rows = [o.tolist() for o in out]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "ranks_seen": seen, "frames_per_step": plan["n_total"], "plan": [r[:3] for r in rows],
                          "pinned": [r[3] for r in rows], "pin_rank0": pin,
                          "predicted": predict_scaling(8.25, args.batch, H, W, D)}), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------
# CPU baseline legs
# ------------------------------------------------------------------------------------------------
def _cpu_one_frame(f):
    """occlusion heuristic + VPP (rnd) + rSGM of synthetic frame f with the CPU oracle; returns (seconds, disparity)."""
    import oracle
    import synth
    fr = synth.make_frame(H, W, D, P_HINTS, seed=1234, frame=f)
    t0 = time.perf_counter()
    oracle.init_rand(1 + f)
    occ = None if _CPU_NO_OCC else oracle.occlusion_heuristic(fr["hints"])[1]
    lv, rv = oracle.vpp(fr["left"], fr["right"], fr["hints"], g_occ=occ)
    disp = oracle.compute_rsgm(fr["left"], lv, rv, dmax=D, subpixel=True)
    return time.perf_counter() - t0, disp


_CPU_NO_OCC = False


def _cpu_worker(f):
    return _cpu_one_frame(f)[0]


def _oracle_flags():
    try:
        with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
            for line in f:
                if line.startswith("CFLAGS"):
                    return line.split("=", 1)[1].strip()
    except OSError:
        pass
    return "?"


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
                sample=f"{n_frames} full {H}x{W}x{D} frames (occlusion heuristic + VPP rnd + rSGM), oracle/liboracle.so "
                       f"gcc {_oracle_flags().split(' -fPIC')[0]}, 1 thread, {t_tot / n_frames:.2f} s/frame, "
                       f"host has {os.cpu_count()} cpus; the port's rSGM is scalar C, about an order of magnitude slower than the "
                       "SSE rSGM of the literature (SURVEY section 6; `cpu_baseline_simd` is the same port with AVX2 aggregation, WTA and median): the GPU/CPU ratio is not a kernel-quality figure.  Calibration "
                       "against the real reference (profiles/r05_cpu_calibration.json, tools/calibrate_cpu.py): the VPP half of the port "
                       "takes 0.49x the time of the reference's Cython scan (5.4 vs 10.9 ms per frame, bit-identical); the rSGM half "
                       "(> 99 % of the time) cannot be calibrated, pyrSGM is not in the reference's tree")
    return base, epe


def cpu_baseline_simd(n_frames, gpu_out=None):
    """The same port with the aggregation -- three quarters of the scalar port's time -- on AVX2 (oracle/rsgm_oracle.c,
    aggregate_paths_avx2: adds_epu16 / min_epu16 / minpos, bit-equal to the scalar function): the KIND of code the reference's
    natives are (SSE, one thread: rsgm.py:44,61).  Still a port, still one thread; the scalar leg stays `cpu_baseline` and the
    checker."""
    import oracle
    oracle.set_simd(True)
    try:
        t_tot, epe = 0.0, None
        for f in range(n_frames):
            dt, disp = _cpu_one_frame(f)
            t_tot += dt
            if gpu_out is not None and f < gpu_out.shape[0]:
                e = float(abs(disp - gpu_out[f]).mean())
                epe = e if epe is None else max(epe, e)
    finally:
        oracle.set_simd(False)
    return dict(value=n_frames * H * W * D / t_tot / 1e6, unit="Mdisparities/s", cores=1, kind="port", s_per_frame=round(t_tot / n_frames, 3),
                epe_vs_gpu=epe, flags=_oracle_flags().split(' -fPIC')[0],
                sample=f"{n_frames} full {H}x{W}x{D} frames, the oracle with its AVX2 twins: aggregation as two raster scans of four paths "
                       "(16 disparities per vector), left / right WTA (16 pixels per vector), 3x3 median; cost volumes kept between frames; the "
                       "census, the cost (popcnt) and the glue stages stay scalar C; 1 thread")


def _cgroup_cpu_quota():
    """CPUs' worth of time this process tree may use per period (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _usable_cores():
    """(processes to start, logical cpus, why): the physical cores this process may really run on at once -- affinity mask, the
    cgroup's CPU quota (the GPU boxes of the pool give a job 16 CPUs' worth of a 128-core host: more processes than that only
    time-slice), and memory (a process holds the oracle's cost and path volumes, ~0.5 GB)."""
    logical = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or logical
        mem_procs = int(psutil.virtual_memory().available / (0.75 * 2 ** 30))
    except Exception:  # noqa: BLE001
        phys, mem_procs = logical, logical
    try:
        phys = min(phys, len(os.sched_getaffinity(0)))
    except Exception:  # noqa: BLE001
        pass
    quota = _cgroup_cpu_quota()
    n = max(1, min(phys, mem_procs, int(quota) if quota and quota >= 1 else phys))
    why = (f"{phys} physical cores in the affinity mask of {logical} logical cpus, cgroup quota "
           f"{'none' if quota is None else f'{quota:g} cpus'}, memory for {mem_procs} processes")
    return n, logical, why


def cpu_baseline_parallel():
    """Same port, one frame per process on every core the job may use (`_usable_cores`)."""
    import multiprocessing as mp
    n, logical, why = _usable_cores()
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(n) as pool:
        per = pool.map(_cpu_worker, list(range(n)), chunksize=1)
    wall = time.perf_counter() - t0
    return dict(value=n * H * W * D / wall / 1e6, unit="Mdisparities/s", cores=n, kind="port",
                sample=f"{n} frames on {n} processes = the cores this job may use ({why}), "
                       f"wall {wall:.1f} s (includes frame synthesis), mean {sum(per) / n:.2f} s/frame/core; the port's "
                       "rSGM is scalar C (about an order of magnitude slower than the SSE rSGM of the literature, SURVEY section 6)")


_SIMD_BARRIER = None


def _cpu_simd_worker(job):
    """k frames with the AVX2 twins in this process; frames are synthesised and one is computed (the kept cost volumes get their
    pages) before every process meets at the barrier; returns (start, end) on the system-wide monotonic clock."""
    f0, k = job
    import oracle
    import synth
    frames = [synth.make_frame(H, W, D, P_HINTS, seed=1234, frame=f0 + i) for i in range(k)]
    oracle.set_simd(True)

    def one(i):
        fr = frames[i]
        oracle.init_rand(1 + f0 + i)
        occ = None if _CPU_NO_OCC else oracle.occlusion_heuristic(fr["hints"])[1]
        lv, rv = oracle.vpp(fr["left"], fr["right"], fr["hints"], g_occ=occ)
        return oracle.compute_rsgm(fr["left"], lv, rv, dmax=D, subpixel=True)

    one(0)
    _SIMD_BARRIER.wait(900)          # a worker that died before it breaks the barrier for everyone: no hang
    t0 = time.perf_counter()
    for i in range(k):
        one(i)
    return t0, time.perf_counter()


def cpu_baseline_simd_parallel(frames_per_process=8):
    """The AVX2 port on every core the job may use at once, one process per core, a few frames each: what the GPU box's host
    share does with the kind of code the reference's natives are."""
    global _SIMD_BARRIER
    import multiprocessing as mp
    n, logical, why = _usable_cores()
    k = frames_per_process
    ctx = mp.get_context("fork")
    _SIMD_BARRIER = ctx.Barrier(n)
    try:
        with ctx.Pool(n) as pool:
            spans = pool.map(_cpu_simd_worker, [(i * k, k) for i in range(n)], chunksize=1)
    finally:
        _SIMD_BARRIER = None
    wall = max(e for _, e in spans) - min(b for b, _ in spans)
    per = sum(e - b for b, e in spans) / (n * k)
    return dict(value=n * k * H * W * D / wall / 1e6, unit="Mdisparities/s", cores=n, kind="port", s_per_frame_per_core=round(per, 3),
                sample=f"{n * k} frames: {k} on each of {n} processes = the cores this job may use ({why}), the oracle with its "
                       f"AVX2 twins, all processes released together after synthesising their frames and one untimed frame; wall {wall:.2f} s "
                       "from the first start to the last end")


def kernel_source_sha():
    """Identity of the HIP sources the PMC profile belongs to (profiles go stale when a kernel changes)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vppstereo_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def load_pmc():
    """The committed PMC profile, or None when it was taken from other kernel sources."""
    try:
        with open(PMC_PROFILE) as f:
            pmc = json.load(f)
    except Exception:
        return None
    return pmc if pmc.get("kernel_source_sha") == kernel_source_sha() else None


def pmc_view(agg_frames, kernel_ms, kernel="sgm_paths_kernel"):
    """PMC numbers of the aggregation kernel from the committed profile, or None when the profile was taken from
    other kernel sources / another batch size."""
    try:
        with open(PMC_PROFILE) as f:
            pmc = json.load(f)
    except Exception:
        return None
    if pmc.get("kernel_source_sha") != kernel_source_sha() or pmc.get("batch") != agg_frames or \
            (pmc.get("H"), pmc.get("W"), pmc.get("D")) != (H, W, D):
        return None
    k = pmc.get("kernels", {}).get(kernel)
    if not k:
        return None
    out = dict(k)
    out["profile"] = os.path.relpath(PMC_PROFILE, ROOT)
    out["profile_commit"] = pmc.get("commit")
    if "hbm_GB_per_launch" in k and kernel_ms > 0:
        out["hbm_frac_of_peak"] = round(k["hbm_GB_per_launch"] / (kernel_ms * 1e-3) / HBM_PEAK_GBS, 4)
    return out


def other_config(eng, torch, synth, name, h, w, d, p, batch, steps=4, warmup=2, tensors=None):
    """One of the other BASELINE.json configurations, same loop as the headline (mask, overlap, inputs resident), bounded:
    `batch` distinct synthetic scenes, a few steps.  Parity at these sizes: tests/test_gpu_fullsize.py, test_gpu_headline.py.
    `tensors`: resident (left, right, hints) to take the first `batch` frames of instead of synthesising new scenes."""
    import numpy as np
    dev = eng.device
    if tensors is not None:
        left, right, hints = (t[:batch].contiguous() for t in tensors)
    else:
        b = synth.make_batch(batch, h, w, d, p, seed=4321)
        left, right, hints = (torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("left", "right", "hints"))
    occ = torch.empty((batch, h, w), dtype=torch.uint8, device=dev)
    outs = [torch.empty((batch, h, w), dtype=torch.float32, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    ev.record()
    torch.cuda.synchronize()

    def one(k):
        eng.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", occ_out=occ, out=outs[k & 1], seed=1,
                     rsgm_kw=dict(dmax=d, subpixel=1), inputs_ready=ev)

    for k in range(warmup):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    eng.status()
    mdisp = batch * h * w * d / ms / 1e3
    return {"config": name, "H": h, "W": w, "D": d, "hint_density": p, "frames_per_step": batch, "steps": steps,
            "ms_per_step": round(ms, 3), "Mdisparities_per_s": round(mdisp, 1),
            "roofline_frac": round(mdisp * 1e6 * (PATH_BYTES_PER_CELL + PATH_BYTES_PER_PIXEL / d) / 1e9 / HBM_PEAK_GBS, 4),
            "aggregation_layout": eng.uses_vert()}


# ------------------------------------------------------------------------------------------------
def run_rank(args):
    global H, W, D, P_HINTS, _CPU_NO_OCC
    if os.environ.get("BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_DEBUG"]), exit=True)
    if args.shape:
        H, W, D, P_HINTS = int(args.shape[0]), int(args.shape[1]), int(args.shape[2]), float(args.shape[3])
    _CPU_NO_OCC = args.no_occ
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from vppstereo_amd import dist as vdist
    from vppstereo_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher decides; reporting n_gpus={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)  # dry runs: several ranks on one GPU
    torch.cuda.set_device(dev_index)
    affinity = pin_to_gpu_numa(torch, dev_index) if world > 1 else None
    backend = args.backend or ("nccl" if world <= ndev else "gloo")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", dev_index)
    eng = Engine(dev_index)
    # consecutive steps are independent batches whose inputs are resident: the front stage of the next step may run on
    # a second stream under the tail of this one (ignored under --graph, where a step is one captured graph)
    eng.set_pipeline(not args.no_pipeline)
    if args.graph:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))  # stream capture needs a non-default stream
        eng.set_graph_mode(True)

    B = args.batch
    plan = frame_plan(B, rank, world)
    n_total, lo, seed0 = plan["n_total"], plan["lo"], plan["seed0"]
    # synthetic frames: every frame of the batch is its own scene (generation is host-side numpy, ~0.4 s per frame)
    n_unique = B
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
    # --graph replays one captured call: a single output buffer keeps its arguments identical from step to step
    n_out = 1 if args.graph else 2
    outs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(n_out)]
    occ_buf = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    pending = [None] * n_out  # gathers in flight, one per output buffer
    nstep = [0]

    # the inputs are resident: everything uploaded above is complete once this event is (it lets the pipelined front stage
    # of a step start under the previous step's tail instead of waiting for the whole launch stream: vppx_inputs_ready_event)
    torch.cuda.synchronize()
    ev_ready = torch.cuda.Event()
    ev_ready.record()
    torch.cuda.synchronize()

    def local_step(buf, use_occ=True, lft=left, rgt=right, hnt=hints, occb=occ_buf):
        # test.py:154 (--maskocc) + :158-225 in one call: mask -> VPP -> rSGM; the mask is delivered to the caller as well
        eng.vpp_rsgm(lft, rgt, hnt, g_occ="occlusion_heuristic" if use_occ else None, occ_out=occb if use_occ else None,
                     out=buf, seed=seed0, rsgm_kw=dict(dmax=D, subpixel=1), inputs_ready=ev_ready)
        return buf

    use_occ = not args.no_occ

    def step():
        # the gather of step k (to rank 0, asynchronous) overlaps the kernels of step k+1, which write the
        # other output buffer; a buffer is reused only after its previous gather has completed
        k = nstep[0] % n_out
        nstep[0] += 1
        if pending[k] is not None:
            t0 = time.perf_counter()
            pending[k].wait()
            gather_wait[0] += time.perf_counter() - t0
            pending[k] = None
        local_step(outs[k], use_occ)
        if world > 1 and not args.no_gather:
            pending[k] = vdist.gather_disparities_async(outs[k], n_total, dst=0)  # every rank calls step() equally often

    gather_wait = [0.0]   # host seconds the step loop spent waiting for a gather (the exposed part of the exchange)

    def drain():
        for k in range(n_out):
            if pending[k] is not None:
                t0 = time.perf_counter()
                pending[k].wait()
                gather_wait[0] += time.perf_counter() - t0
                pending[k] = None

    if world > 1 and not args.no_gather:
        # RCCL creates its send/recv channels on first use: do that outside the timed region even with --warmup 0.
        # A gather that does not work is an ERROR (non-zero exit status on every rank): a line without the exchange would
        # look like a multi-GPU result and not be one.  `--no-gather` is the explicit way to time the shards alone.
        ok, why = 1, ""
        try:
            vdist.gather_disparities_async(outs[0], n_total, dst=0).wait()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {e}"
            print(f"[bench] rank {rank}: result gather failed ({why})", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[bench] rank {rank}: all_reduce failed ({type(e).__name__}: {e})", file=sys.stderr)
        if ok == 0:
            raise SystemExit(f"[bench] rank {rank}: the result gather to rank 0 does not work on this node ({why or 'another rank failed'}); "
                             "fix the process group or pass --no-gather to time the shards without the exchange")

    def timed(n_steps, fn):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            fn()
        drain()  # the last gathers complete inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        step()
    drain()
    eng.agg_kernel_ms(0)  # reset: the hipEvent pairs around the aggregation launches of the timed steps only
    eng.we_kernel_ms(0)
    gather_wait[0] = 0.0
    dt_local = timed(args.steps, step)
    gather_ms_local = gather_wait[0] / args.steps * 1e3
    eng.status()   # raises if a fused aggregation launch of the timed region lost its lock step (its disparities would be void)
    dt = dt_local
    rank_ms = [dt_local / args.steps * 1e3]
    rank_gather_ms = [gather_ms_local]
    rank_numa = [affinity.get("numa_node") if affinity and affinity.get("pinned") else None]
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([dt_local, gather_ms_local, float(rank_numa[0]) if rank_numa[0] is not None else -1.0], dtype=torch.float64, device=cdev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_ms = [float(x[0].item()) / args.steps * 1e3 for x in allt]
        rank_gather_ms = [float(x[1].item()) for x in allt]
        rank_numa = [int(x[2].item()) if x[2].item() >= 0 else None for x in allt]
        dt = max(float(x[0].item()) for x in allt)
    if args.graph:
        assert eng.graph_replays() > 0, "--graph was given but no call was served by a graph replay"
    # power steady state: the timed region is a fraction of a second on a chip that clocks by its power budget; the same loop
    # for `--sustained-steps` further steps (every rank: `timed` has barriers), reported next to the K-step figure
    ms_sustained = None
    if args.sustained_steps > 0:
        ms_sustained = timed(args.sustained_steps, step) / args.sustained_steps * 1e3
        if world > 1:
            t = torch.tensor([ms_sustained], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_sustained = float(t.item())
        eng.status()
    # the same steps with the cross-step overlap off (every rank: `timed` has barriers)
    ms_unpipelined = None
    if not (args.no_pipeline or args.graph):
        eng.set_pipeline(False)
        n_unp = max(2, min(5, args.steps))
        step()
        drain()
        ms_unpipelined = timed(n_unp, step) / n_unp * 1e3
        if world > 1:
            t = torch.tensor([ms_unpipelined], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_unpipelined = float(t.item())
        eng.status()
        eng.set_pipeline(True)

    ms_per_step = dt / args.steps * 1e3
    value = n_total * H * W * D * args.steps / dt / 1e6

    result = None
    if rank == 0:
        # ---- dominant kernel, measured live with hipEvents on its own launch stream ----------
        # average over the launches made INSIDE the timed region (event pairs on the launch stream, ring of 64)
        agg_ms, agg_n = eng.agg_kernel_ms(args.steps)
        we_ms_in_step, we_n = eng.we_kernel_ms(args.steps)   # the W/E launch of the fused layout inside the timed steps
        layout = eng.uses_vert()   # 0: one launch, eight line-parallel paths; 3: W/E launch + the fused vertical kernel
        fused = layout == 3
        # same kernel re-launched back to back (fused layout: its part of the aggregation only)
        agg_ms_b2b = eng.time_aggregate_part(2, max(3, min(10, args.steps))) if fused else eng.time_aggregate(iters=max(3, min(10, args.steps)))
        if agg_n == 0 or agg_ms <= 0:   # graph replays carry no event pairs
            agg_ms = agg_ms_b2b
        agg_frames = eng.time_aggregate_frames()  # frames per launch (the batch is split over sub-streams)
        cells_launch = agg_frames * H * W * D     # SURVEY 8d counts unpadded cells
        launches_per_step = max(1, -(-B // agg_frames)) if agg_frames else 1
        # SURVEY 8d prices the 8-path aggregation at 10 B/cell; the fused vertical kernel carries 6 of the 8 paths
        agg_bytes = AGG_BYTES_PER_CELL * (6.0 / 8.0 if fused else 1.0)
        wide = fused and eng.fused_pixels_per_wave() == 16
        dom_kernel = ("sgm_vert4_kernel" if wide else "sgm_vert3_kernel") if fused else "sgm_paths_kernel"
        achieved = cells_launch * agg_bytes / (agg_ms * 1e-3) / 1e9
        pmc = pmc_view(agg_frames, agg_ms, dom_kernel)
        we_ms = eng.time_aggregate_part(1, 3) if fused else None
        # the same step without the occlusion mask (the reference's default, test.py --maskocc off): rank-local, no collectives
        other = None
        if world == 1:
            n2 = max(2, min(5, args.steps))
            local_step(outs[0], not use_occ)
            dt2 = timed(n2, lambda: local_step(outs[0], not use_occ))
            other = round(dt2 / n2 * 1e3, 3)
        # ---- the literal cfg 2: one pair per call ------------------------------------------------
        l1, r1, h1, o1 = left[:1].contiguous(), right[:1].contiguous(), hints[:1].contiguous(), occ_buf[:1]
        out1 = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        for _ in range(3):
            local_step(out1, use_occ, l1, r1, h1, o1)
        torch.cuda.synchronize()
        n1 = 20
        t0 = time.perf_counter()
        for _ in range(n1):
            local_step(out1, use_occ, l1, r1, h1, o1)
        torch.cuda.synchronize()
        b1_ms = (time.perf_counter() - t0) / n1 * 1e3
        eng.enable_stage_timing(True)
        for _ in range(2):  # first pass sizes the un-split workspace; report the second
            local_step(outs[0], use_occ)    # rank-0-only section: no collectives here
            torch.cuda.synchronize()
        stages = eng.stage_ms()
        eng.enable_stage_timing(False)
        pipeline_gbs = value * 1e6 * (PATH_BYTES_PER_CELL + PATH_BYTES_PER_PIXEL / D) / 1e9
        # ---- the dominant kernel as it is: issue-bound, so its own bound is the VALU issue time, not a byte count ----
        pmc_all = load_pmc()
        pmc_ok = bool(pmc_all) and pmc_all.get("batch") == agg_frames and (pmc_all.get("H"), pmc_all.get("W"), pmc_all.get("D")) == (H, W, D)
        step_traffic = round(pmc_all["hbm_GB_per_step"], 3) if pmc_ok and pmc_all.get("hbm_GB_per_step") else None
        valu = None
        if pmc and pmc.get("SQ_INSTS_VALU"):
            # shader clock of THIS kernel under load: GRBM_GUI_ACTIVE (shader-clock cycles per XCD, summed over the 8 XCDs)
            # / 8 / the kernel's duration, both from the SQ pass of the committed profile
            clk = pmc.get("sclk_mhz_profiled")
            if clk:
                floor_ms = pmc["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / (N_SIMD * clk * 1e6) * 1e3
                valu = {"SQ_INSTS_VALU_per_launch": pmc["SQ_INSTS_VALU"], "cycles_per_inst": VALU_CYCLES_PER_INST,
                        "simds": N_SIMD, "sclk_mhz": clk, "sclk_source": "profile run: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration",
                        "issue_floor_ms": round(floor_ms, 3), "frac_of_issue_floor": round(floor_ms / agg_ms, 4),
                        "wave_cycles_active_issuewait_memwait": [pmc.get("sq_active_inst_any_frac_of_wave_cycles"),
                                                                 pmc.get("sq_wait_inst_any_frac_of_wave_cycles"),
                                                                 pmc.get("sq_wait_any_frac_of_wave_cycles")]}
        dom = {"kernel": (f"{dom_kernel} (N, NW, NE and S, SW, SE fused three at a time: 6 of the 8 aggregation paths; "
                          f"{16 if wide else 8} pixels per wave)" if fused else "sgm_paths_kernel (8-path aggregation)"),
               "limited_by": "valu issue",
               "kernel_ms": round(agg_ms, 4), "kernel_launches_timed": agg_n, "kernel_ms_back_to_back": round(agg_ms_b2b, 4),
               "launches_per_step": launches_per_step, "share_of_step": round(agg_ms * launches_per_step / ms_per_step, 3),
               "valu": valu,
               "traffic_unit": "GB per launch (PMC)",
               "hbm_frac_of_peak": pmc.get("hbm_frac_of_peak") if pmc else None,
               "algorithmic_bytes": f"{agg_bytes} B/cell = SURVEY 8d's 10 B/cell of the 8-path aggregation, pro rata for the paths this launch "
                             f"carries: {round(cells_launch * agg_bytes / 1e9, 3)} GB per launch / kernel_ms vs 8 TB/s (a byte-count "
                             "convention, not a utilisation: see valu and hbm_frac_of_peak)",
               "other_aggregation_launch_ms": ({("sgm_we12_kernel (W, E)" if D in (128, 192, 256) else "sgm_paths_kernel (W, E)"): {"in_step": round(we_ms_in_step, 4) if we_n else None,
                                                                            "launches_timed": we_n, "back_to_back": round(we_ms, 4)}}
                                               if fused else None),
               "frames_per_launch": agg_frames, "cells_per_launch": cells_launch,
               "pmc": pmc}
        # ---- the three big kernels of a part side by side: the byte-count convention (SURVEY 8d's share of the 16 B/cell), the
        # real HBM utilisation and the VALU issue floor (both from the committed PMC profile of THESE kernel sources, else null)
        def kernel_row(pmc_name, ms, bytes_per_cell, what):
            if not ms or ms <= 0:
                return None
            row = {"what": what, "ms_per_launch": round(ms, 4), "launches_per_step": launches_per_step,
                   "convention_bytes_per_cell": bytes_per_cell,
                   "convention_frac": round(cells_launch * bytes_per_cell / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "hbm_frac_of_peak": None, "frac_of_issue_floor": None}
            pk = pmc_view(agg_frames, ms, pmc_name)
            if pk:
                row["hbm_frac_of_peak"] = pk.get("hbm_frac_of_peak")
                clk = pk.get("sclk_mhz_profiled")
                if pk.get("SQ_INSTS_VALU") and clk:
                    row["frac_of_issue_floor"] = round(pk["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / (N_SIMD * clk * 1e6) * 1e3 / ms, 4)
                    row["SQ_INSTS_VALU_per_launch"] = pk["SQ_INSTS_VALU"]
            return row
        kernels = None
        if fused:
            sum_ms = stages.get("sum_wta_left", 0.0) / launches_per_step
            we_name = "sgm_we12_kernel" if D in (128, 192, 256) else "sgm_paths_kernel"
            kernels = {dom_kernel: kernel_row(dom_kernel, agg_ms, agg_bytes, "N, NW, NE + S, SW, SE (in-step hipEvent pairs)"),
                       we_name: kernel_row(we_name, we_ms_in_step if we_n else we_ms, AGG_BYTES_PER_CELL * 2.0 / 8.0, "W, E (in-step hipEvent pairs)"),
                       ("sum_wta_trap_kernel" if D == 256 else "sum_wta_lr_kernel"):
                           kernel_row("sum_wta_trap_kernel" if D == 256 else "sum_wta_lr_kernel", sum_ms, 4.0, "sum + left / right WTA + sub-pixel (stage-timed pass, overlap off)")}
        occ_txt = "occlusion heuristic + " if use_occ else ""
        result = {
            "metric": f"Mdisparities/s (HxWxD / s) VPP+rSGM at {H}x{W}xD={D}",
            "value": round(value, 1), "unit": "Mdisparities/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_unpipelined": round(ms_unpipelined, 3) if ms_unpipelined else None,
            "ms_per_step_sustained": round(ms_sustained, 3) if ms_sustained else None, "sustained_steps": args.sustained_steps if ms_sustained else None,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u16",
            "data": "synthetic (uniform random u8)" if args.uniform_random else "synthetic",
            "config": {"workload": f"{H}x{W} RGB pair, {100 * P_HINTS:g}% hints, {occ_txt}VPP(rnd, wsize 3)+rSGM D={D} subpixel, "
                                   f"{B} frames/GPU/step resident in HBM", "frames_per_step": n_total,
                       "H": H, "W": W, "D": D, "hint_density": P_HINTS, "g_occ": "occlusion_heuristic" if use_occ else None,
                       "parts": (f"the library runs a call of {B} frames as {-(-B // agg_frames)} consecutive parts of {agg_frames} frames (one round of "
                                 "the lock-step aggregation kernel each; results do not depend on the split; VPPX_CHUNK=0 keeps whole batches)"
                                 if 0 < agg_frames < B else None),
                       "cross_step_overlap": (None if (args.no_pipeline or args.graph) else
                                              "front stage of step k+1 (occlusion heuristic, VPP, pad+gray, census) on a second "
                                              "stream under the sum/WTA and post kernels of step k; inputs are resident and declared "
                                              "ready (vppx_inputs_ready_event), outputs are delivered in launch-stream order; "
                                              "ms_per_step_unpipelined is the same loop with the overlap off")},
            "ranks": {"launched": args.gpus, "seen": dist.get_world_size() if world > 1 else 1,
                      "backend": (backend if backend != "nccl" else "nccl (RCCL)") if world > 1 else None,
                      "ms_per_step_per_rank": [round(x, 3) for x in rank_ms],
                      "gather_wait_ms_per_step_per_rank": [round(x, 3) for x in rank_gather_ms] if world > 1 else None,
                      "numa_node_per_rank": rank_numa if world > 1 else None,
                      # the curve the first multi-GPU run is to be checked against, from this run's own
                      # one-GPU step time when N = 1, else from the slowest non-root rank
                      "predicted": predict_scaling(ms_per_step if world == 1 else (sorted(rank_ms)[0] if rank_ms else ms_per_step), B, H, W, D)},
            # The dominant kernel (bench contract): SURVEY 8d's algorithmic bytes of the launch -- 10 B/cell for the 8-path
            # aggregation, pro rata for the paths the launch carries -- over its average duration inside the timed steps
            # (hipEvent pairs on the launch stream), against the HBM3E peak; `traffic` = the launch's real HBM bytes (PMC).
            # It is a byte-count convention: the kernel is bound by instruction issue (`valu`), `hbm_frac_of_peak` is its
            # utilisation.  The whole step against the same peak (north_star's ">= 60 %") is `pipeline_roofline`.
            # `bound` names the real limiter: every fused kernel here is bound by VALU instruction issue, so `frac` (the contract's
            # A/P against the HBM peak) is a byte-count convention, `frac_of_issue_floor` the kernel's distance from ITS bound
            # and `hbm_frac_of_peak` its real memory utilisation.
            "roofline": dict({"bound": "valu" if fused else "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 4),
                              "frac_is": "algorithmic bytes (SURVEY 8d) / kernel time vs the HBM peak: a byte-count convention for an issue-bound kernel",
                              "frac_of_issue_floor": valu["frac_of_issue_floor"] if valu else None,
                              "traffic": pmc["hbm_GB_per_launch"] if pmc and "hbm_GB_per_launch" in pmc else None,
                              "kernels": kernels}, **dom),
            "pipeline_roofline": {"what": "whole step at SURVEY 8d's algorithmic bytes, H*W*(16*D + 50) B per frame, vs the HBM3E peak "
                                          "(the figure north_star's 60 % is about; it prices the REFERENCE's dataflow, the build moves less: traffic)",
                                  "bytes_per_cell": PATH_BYTES_PER_CELL, "bytes_per_pixel": PATH_BYTES_PER_PIXEL,
                                  "achieved_GBps": round(pipeline_gbs, 1), "peak": HBM_PEAK_GBS, "frac": round(pipeline_gbs / HBM_PEAK_GBS, 4),
                                  "frac_unpipelined": (round(pipeline_gbs * ms_per_step / ms_unpipelined / HBM_PEAK_GBS, 4) if ms_unpipelined else None),
                                  "algorithmic_GB_per_step": round(n_total / world * H * W * (PATH_BYTES_PER_CELL * D + PATH_BYTES_PER_PIXEL) / 1e9, 3),
                                  "traffic": step_traffic,
                                  "traffic_unit": "GB of real HBM traffic per step and GPU, all kernels (PMC FETCH_SIZE x2 + WRITE_SIZE, separate "
                                                  "rocprofv3 passes; null unless the committed profile was taken from these kernel sources)",
                                  "hbm_frac_of_peak": (round(step_traffic / (ms_per_step * 1e-3) / HBM_PEAK_GBS, 4) if step_traffic else None)},
            ("ms_per_step_without_g_occ" if use_occ else "ms_per_step_with_g_occ"): other,
            "b1": {"ms_per_frame": round(b1_ms, 4), "Mdisparities_per_s": round(H * W * D / b1_ms / 1e3, 1),
                   "what": "one frame per call (B=1), host-paced loop of 20 calls, same stages as the step"
                           + ("" if (args.no_pipeline or args.graph) else "; consecutive calls overlap like the steps (cross_step_overlap)")},
            "stage_ms": {k: round(v, 3) for k, v in stages.items()},
            "kernel_source_sha": kernel_source_sha(),
            "device": eng.ctx.device_name,
        }
        if args.graph:
            result["graph_replays"] = eng.graph_replays()
        if world > 1:
            result["config"]["result_gather"] = "off (--no-gather)" if args.no_gather else "async gather to rank 0, overlapped"
        if world == 1 and args.cpu_frames > 0:
            gpu_out = local_step(outs[0], use_occ)[: min(args.cpu_frames, n_unique)].cpu().numpy()
            result["cpu_baseline"], epe = cpu_baseline(min(args.cpu_frames, n_unique), gpu_out)
            result["epe_vs_cpu_oracle"] = epe      # mean |disp_gpu - disp_cpu|, worst frame (0.0 = bit-equal)
            result["speedup_vs_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
            result["cpu_baseline_simd"] = cpu_baseline_simd(min(args.cpu_frames, n_unique), gpu_out)
            if not args.no_cpu_parallel:
                result["cpu_baseline_all_cores"] = cpu_baseline_parallel()
                result["cpu_baseline_simd_all_cores"] = cpu_baseline_simd_parallel()
                # the ratio worth quoting: one GPU against every host core the job may use, running the reference's kind of code
                result["speedup_vs_cpu_simd_all_cores"] = round(value / result["cpu_baseline_simd_all_cores"]["value"], 1)
        else:
            result["cpu_baseline"] = None
        if world == 1 and not args.no_other_configs and not args.shape and not args.uniform_random and not args.graph:
            # the other BASELINE.json configurations on this one GPU (their multi-GPU form shards frames: section 7 of DESIGN.md)
            q3 = eng.batch_quantum(375, 1242, 192)
            cfg3_q = 3 * q3 if 0 < q3 <= 16 else 0
            result["config"]["batch_quantum"] = {"540x960x192": eng.batch_quantum(H, W, D), "375x1242x192": q3, "1536x2048x256": eng.batch_quantum(1536, 2048, 256)}
            b3 = synth.make_batch(max(32, cfg3_q), 375, 1242, 192, 0.05, seed=4321)
            t3 = tuple(torch.from_numpy(np.ascontiguousarray(b3[k])).to(dev) for k in ("left", "right", "hints"))
            cfg3 = [other_config(eng, torch, synth, "cfg3: KITTI-sized 375x1242 stream, 5% hints, D=192, 16 frames per step (BASELINE's 32 frames "
                                 "over 2 GPUs: the per-GPU batch)", 375, 1242, 192, 0.05, 16, tensors=t3),
                    other_config(eng, torch, synth, "cfg3: the same stream, 32 frames per step on one GPU", 375, 1242, 192, 0.05, 32, tensors=t3)]
            if cfg3_q and cfg3_q not in (16, 32):
                # frames per step = three batch quanta of this width (vppx_batch_quantum: whole rounds of the lock-step kernel)
                cfg3.append(other_config(eng, torch, synth, "cfg3: the same stream, %d frames per step (three batch quanta of this width)" % cfg3_q,
                                         375, 1242, 192, 0.05, cfg3_q, tensors=t3))
            del t3, b3
            result["other_configs"] = cfg3 + [
                other_config(eng, torch, synth, "cfg5: 1536x2048 indoor pairs, 1% hints, D=256, 8 frames per step", 1536, 2048, 256, 0.01, 8, steps=4, warmup=2),
                {"config": "cfg2 literal: one 540x960 pair per call, D=192", "ms_per_frame": round(b1_ms, 4),
                 "Mdisparities_per_s": round(H * W * D / b1_ms / 1e3, 1),
                 "roofline_frac": round(H * W * D / b1_ms / 1e3 * 1e6 * (PATH_BYTES_PER_CELL + PATH_BYTES_PER_PIXEL / D) / 1e9 / HBM_PEAK_GBS, 4)},
            ]
            # latency / throughput curve of the headline shape: frames per call from the reference's operating point (one pair,
            # test.py:293) to the headline batch, same loop (mask, overlap, inputs resident), the first b frames of the batch
            sweep = []
            for bsz in (1, 2, 4, 8, 16):
                if bsz >= B:
                    break
                r = other_config(eng, torch, synth, f"{bsz} frames per call", H, W, D, P_HINTS, bsz, steps=max(4, min(24, 64 // bsz)), warmup=3,
                                 tensors=(left, right, hints))
                sweep.append({"frames_per_call": bsz, "ms_per_call": r["ms_per_step"], "ms_per_frame": round(r["ms_per_step"] / bsz, 4),
                              "Mdisparities_per_s": r["Mdisparities_per_s"], "roofline_frac": r["roofline_frac"],
                              "aggregation_layout": r["aggregation_layout"]})
            sweep.append({"frames_per_call": B, "ms_per_call": round(ms_per_step, 3), "ms_per_frame": round(ms_per_step / B, 4),
                          "Mdisparities_per_s": round(value, 1), "roofline_frac": round(pipeline_gbs / HBM_PEAK_GBS, 4), "aggregation_layout": layout})
            result["batch_sweep"] = sweep
            # ---- the same path for a caller that holds ONE pageable numpy frame at a time (test.py:291-311, then :154-225 per frame):
            # vppstereo_amd.pipeline.FrameStream (vppx_fstream_*).  PCIe inclusive, host copies inclusive: never `value`.
            try:
                from tools import host_stream as hs
                result["host_stream"] = {
                    "what": "pipeline.FrameStream: frames pushed one at a time from pageable numpy arrays, disparities popped one at a time into "
                            "fresh numpy arrays, in input order; occlusion mask on the way; batches of one lock-step round, a ring of three batches (results leave through a copy-out kernel of 8 workgroups behind the NEXT batch's aggregation); ms per frame "
                            "next to pipeline.run_frame (one synchronous call per frame) and the PCIe floor of the bytes that cross",
                    "runs": [hs.measure(H, W, D, P_HINTS, 384), hs.measure(375, 1242, 192, 0.05, 384)]}
            except Exception as e:  # noqa: BLE001  (a measurement next to the headline: its failure must not lose the line)
                result["host_stream"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, argv))
    if args.spawn_check:
        sys.exit(spawn_check(args))
    return run_rank(args)


if __name__ == "__main__":
    main()
