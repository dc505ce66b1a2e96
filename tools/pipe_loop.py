"""tools/pipe_loop.py [steps [B H W D [p]]]: nothing but the pipelined headline loop (B = 32, 540x960x192, mask, inputs declared ready), for
kernel traces: `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 tools/pipe_loop.py 12`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, H, W, D = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (32, 540, 960, 192)
P = float(sys.argv[6]) if len(sys.argv) > 6 else 0.03
eng = Engine()
eng.set_pipeline(os.environ.get("NO_PIPE") is None)
nu = min(B, 8)
b = synth.make_batch(nu, H, W, D, P, seed=1234)
idx = [i % nu for i in range(B)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
outs = [torch.empty((B, H, W), dtype=torch.float32, device=eng.device) for _ in range(2)]
occ = torch.empty((B, H, W), dtype=torch.uint8, device=eng.device)
torch.cuda.synchronize()
ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
for k in range(3):
    eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occ, out=outs[k & 1], rsgm_kw=dict(dmax=D), inputs_ready=ev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(steps):
    eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occ, out=outs[k & 1], rsgm_kw=dict(dmax=D), inputs_ready=ev)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / steps * 1e3, flush=True)
eng.status()
