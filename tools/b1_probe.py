"""tools/b1_probe.py: 40 pipelined single-pair calls (540x960x192, occlusion heuristic + VPP + rSGM, inputs resident) for a
kernel trace of the steady state: rocprofv3 --kernel-trace ... -- python3 tools/b1_probe.py, then tools/b1_sum.py <trace csv>
prints one period (from one 8-path aggregation launch to the next)."""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine
B=1; H,W,D=540,960,192
eng=Engine()
b=synth.make_batch(1,H,W,D,0.03,seed=1234)
l,r,h=(torch.from_numpy(np.ascontiguousarray(b[k])).to(eng.device) for k in ("left","right","hints"))
outs=[torch.empty((B,H,W),dtype=torch.float32,device=eng.device) for _ in range(2)]
occ=torch.empty((B,H,W),dtype=torch.uint8,device=eng.device)
torch.cuda.synchronize()
ev=torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
eng.set_pipeline(True)
for k in range(40):
    eng.vpp_rsgm(l,r,h,g_occ="occlusion_heuristic",occ_out=occ,out=outs[k&1],rsgm_kw=dict(dmax=D),inputs_ready=ev)
torch.cuda.synchronize()
eng.status()
