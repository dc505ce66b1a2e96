"""tools/small_probe.py [B [steps]]: the fused call (occlusion heuristic + VPP + rSGM) WITHOUT the cross-call overlap, so that
every front- and post-stage kernel runs alone on the device; meant to run under `rocprofv3 --kernel-trace --stats`
(tools/small_probe.sh), whose per-kernel averages are then the stand-alone durations of the small kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H, W, D = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (540, 960, 192)
eng = Engine()
nu = min(B, 8)
b = synth.make_batch(nu, H, W, D, float(os.environ.get("PROBE_DENSITY", "0.03")), seed=1234)
idx = [i % nu for i in range(B)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
out = torch.empty((B, H, W), dtype=torch.float32, device=eng.device)
occ = torch.empty((B, H, W), dtype=torch.uint8, device=eng.device)
eng.set_pipeline(False)
for k in range(steps):
    eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occ, out=out, rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()
eng.status()
