"""tools/rccl_contention.py [out.json] [steps]: rank 0's load of an 8-rank run, on the ONE GPU a gpurun box has (VERDICT r04
item 3: de-risk the first N > 1 run).  With 8 ranks rank 0 receives 7 shards of B x H x W float32 (66 MB each at 32 x 540 x 960)
per step while its own kernels run -- among them the lock-step kernel, which wants an XCD's block slots to itself.  A fresh
process forms a one-rank `nccl` (RCCL) group and runs the headline loop (32 frames, mask, pipelined) four ways, alternated:
  none      no exchange
  rccl x7   7 x gather_disparities_async(force_collective=True) of a 66 MB shard per step: RCCL's kernels on its own stream
  rccl x1   the same with one shard (what a 2-rank run costs rank 0)
  copy      a 464 MB device-to-device torch copy per step on a side stream (a CU copy kernel: the stand-in for receive kernels)
and reports ms per step, the lock-step kernel's in-step duration, lost lock steps and how long the step loop waited for the
exchange.  NCCL_MAX_NCHANNELS from the environment is echoed (bench.py's launcher sets it for N > 1)."""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else None
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from vppstereo_amd import dist as vdist
    from vppstereo_amd.engine import Engine
    B, H, W, D = 32, 540, 960, 192
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eng = Engine(0)
    eng.set_pipeline(True)
    nu = 8
    b = synth.make_batch(nu, H, W, D, 0.03, seed=1234)
    idx = [i % nu for i in range(B)]
    left, right, hints = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(dev) for k in ("left", "right", "hints"))
    outs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
    occ = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    big_src = torch.empty((7, B, H, W), dtype=torch.float32, device=dev)
    big_dst = torch.empty_like(big_src)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
    vdist.gather_disparities_async(outs[0].zero_(), B, dst=0, force_collective=True).wait()   # channels are created on first use
    torch.cuda.synchronize()
    fails = eng.lib.vppx_lockstep_failures
    fails.restype = __import__("ctypes").c_long

    def loop(mode, n):
        pending = [[], []]
        waited = 0.0
        done_ev = [None, None]
        for s_ in range(n):
            k = s_ & 1
            t0 = time.perf_counter()
            for h in pending[k]:
                h.wait()
            pending[k] = []
            if done_ev[k] is not None:
                torch.cuda.current_stream().wait_event(done_ev[k])
            waited += time.perf_counter() - t0
            eng.vpp_rsgm(left, right, hints, g_occ="occlusion_heuristic", occ_out=occ, out=outs[k], seed=1, rsgm_kw=dict(dmax=D, subpixel=1),
                         inputs_ready=ev)
            if mode.startswith("rccl"):
                for _ in range(int(mode[5:])):
                    pending[k].append(vdist.gather_disparities_async(outs[k], B, dst=0, force_collective=True))
            elif mode == "copy":
                e = torch.cuda.Event(); e.record()
                side.wait_event(e)
                with torch.cuda.stream(side):
                    big_dst.copy_(big_src, non_blocking=True)
                    d = torch.cuda.Event(); d.record()
                done_ev[k] = d
        for k in (0, 1):
            for h in pending[k]:
                h.wait()
        return waited

    res = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "shape": [B, H, W, D], "steps": steps,
           "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), "runs": []}
    for rnd in range(3):
        for mode in ("none", "rccl_7", "rccl_1", "copy"):
            loop(mode, 3)
            torch.cuda.synchronize()
            eng.agg_kernel_ms(0); eng.we_kernel_ms(0)
            f0 = int(fails(eng.ctx.handle))
            t0 = time.perf_counter()
            waited = loop(mode, steps)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            lost = int(fails(eng.ctx.handle)) - f0
            try:
                eng.status()
                st = "ok"
            except Exception as e:  # noqa: BLE001
                st = str(e)[:120]
            res["runs"].append({"round": rnd, "exchange": mode, "ms_per_step": round(ms, 3), "vert4_in_step_ms": round(eng.agg_kernel_ms(2 * steps)[0], 4),
                                "we_in_step_ms": round(eng.we_kernel_ms(2 * steps)[0], 4), "lost_lock_steps": lost, "status": st,
                                "host_wait_for_exchange_ms_per_step": round(waited / steps * 1e3, 3)})
            print(res["runs"][-1], flush=True)
    base = [r["ms_per_step"] for r in res["runs"] if r["exchange"] == "none"]
    for mode in ("rccl_7", "rccl_1", "copy"):
        v = [r["ms_per_step"] for r in res["runs"] if r["exchange"] == mode]
        res[f"slowdown_{mode}"] = round(sum(v) / len(v) / (sum(base) / len(base)), 4)
    res["lost_lock_steps_total"] = sum(r["lost_lock_steps"] for r in res["runs"])
    dist.destroy_process_group()
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
