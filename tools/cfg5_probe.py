"""tools/cfg5_probe.py: pipelined steps of 8 frames of 1536x2048x256 (cfg 5 per GPU) with the stage times of one un-overlapped step."""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine
B=8; H,W,D=1536,2048,256
eng=Engine()
b=synth.make_batch(4,H,W,D,0.01,seed=1234)
idx=[i%4 for i in range(B)]
l,r,h=(torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left","right","hints"))
outs=[torch.empty((B,H,W),dtype=torch.float32,device=eng.device) for _ in range(2)]
torch.cuda.synchronize()
ev=torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
eng.set_pipeline(True)
for k in range(8):
    eng.vpp_rsgm(l,r,h,g_occ="occlusion_heuristic",out=outs[k&1],rsgm_kw=dict(dmax=D),inputs_ready=ev)
torch.cuda.synchronize()
eng.status()
