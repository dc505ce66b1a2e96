"""tools/agg_probe.py [B [H W D]]: aggregation stage timings at the headline shape for the experiment knobs of the environment
(VPPX_WE_OVERLAP, VPPX_V3_SPIN_LIMIT=1 = fused kernel without any neighbour wait: its compute-only time, results void)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W, D = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (540, 960, 192)
eng = Engine()
b = synth.make_batch(min(B, 8), H, W, D, 0.03, seed=1234)
idx = [i % min(B, 8) for i in range(B)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
out = torch.empty((B, H, W), dtype=torch.float32, device=eng.device)
for _ in range(2):
    eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", out=out, rsgm_kw=dict(dmax=D))
torch.cuda.synchronize()
res = dict(shape=(B, H, W, D), layout=eng.uses_vert(), agg_all=round(eng.time_aggregate(10), 3), we=round(eng.time_aggregate_part(1, 10), 3),
           vert=round(eng.time_aggregate_part(2, 10), 3))
n = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", out=out, rsgm_kw=dict(dmax=D))
torch.cuda.synchronize()
res["step_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
res["env"] = {k: v for k, v in os.environ.items() if k.startswith("VPPX_")}
try:
    eng.status()
except Exception as e:
    res["status"] = str(e)[:60]
print(res, flush=True)
