"""tools/maxdist_latency.py: one 540x960 frame through the maxDistance colour method (Engine.vpp), ms per call."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth
from vppstereo_amd.engine import Engine
eng = Engine()
dev = eng.device
for B in (1, 8, 32):
    b = synth.make_batch(min(B,4), 540, 960, 192, 0.03, seed=1234)
    idx=[i%min(B,4) for i in range(B)]
    l = torch.from_numpy(np.ascontiguousarray(b["left"][idx])).to(dev); r = torch.from_numpy(np.ascontiguousarray(b["right"][idx])).to(dev)
    h = torch.from_numpy(np.ascontiguousarray(b["hints"][idx])).to(dev)
    for it in range(2):
        torch.cuda.synchronize(); t=time.time()
        eng.vpp(l, r, h, seed=1, method=1, uniform_color=int(os.environ.get("UNI", "0")))
        torch.cuda.synchronize(); dt=time.time()-t
    print("maxDistance B=%d: %.1f ms (%.1f ms/frame)" % (B, dt*1e3, dt*1e3/B))
