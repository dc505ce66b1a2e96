#!/usr/bin/env python3
"""Time the parts of the aggregation stage (experiment helper; GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, synth
from vppstereo_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W, D = 540, 960, 192
b = synth.make_batch(min(B, 2), H, W, D, 0.03)
idx = [i % 2 for i in range(B)]
eng = Engine(0)
dev = eng.device
l = torch.from_numpy(np.ascontiguousarray(b["left"][idx])).to(dev)
r = torch.from_numpy(np.ascontiguousarray(b["right"][idx])).to(dev)
h = torch.from_numpy(np.ascontiguousarray(b["hints"][idx])).to(dev)
eng.vpp_rsgm(l, r, h, rsgm_kw=dict(dmax=D))
torch.cuda.synchronize()
print("B", B, "vert", eng.uses_vert(), "all", round(eng.time_aggregate(3), 3), "horiz", round(eng.time_aggregate_part(1, 3), 3),
      "vert-bands", round(eng.time_aggregate_part(2, 3), 3))
