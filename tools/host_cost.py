"""tools/host_cost.py [B]: host time to enqueue one pipelined vpp_rsgm call (the GPU queue is empty at the start of the
measured burst, so the calls never wait for the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W, D = 540, 960, 192
eng = Engine()
eng.set_pipeline(os.environ.get("NO_PIPE") is None)
b = synth.make_batch(B, H, W, D, 0.03, seed=1)
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k])).to(eng.device) for k in ("left", "right", "hints"))
outs = [torch.empty((B, H, W), dtype=torch.float32, device=eng.device) for _ in range(2)]
occ = torch.empty((B, H, W), dtype=torch.uint8, device=eng.device)
ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
for k in range(5):
    eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occ, out=outs[k & 1], rsgm_kw=dict(dmax=D), inputs_ready=ev)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4):
        eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occ, out=outs[k & 1], rsgm_kw=dict(dmax=D), inputs_ready=ev)
    ts.append((time.perf_counter() - t0) / 4 * 1e3)
    torch.cuda.synchronize()
print("host ms per call (4-call bursts):", [round(t, 3) for t in ts], flush=True)
