"""Probe: software pipeline over consecutive batches -- occlusion heuristic + VPP of batch k+1 on a second stream and
context while rSGM of batch k runs on the first (the patterned pair is double buffered).  Same total work per batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth
from vppstereo_amd.engine import Engine
H, W, D, B = 540, 960, 192, 32
dev = torch.device("cuda", 0)
eng_v, eng_r = Engine(0), Engine(0)
b = synth.make_batch(4, H, W, D, 0.03, seed=1234)
idx = [i % 4 for i in range(B)]
left = torch.from_numpy(np.ascontiguousarray(b["left"][idx])).to(dev)
right = torch.from_numpy(np.ascontiguousarray(b["right"][idx])).to(dev)
hints = torch.from_numpy(np.ascontiguousarray(b["hints"][idx])).to(dev)
out = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
occ = [torch.empty((B, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
s_v, s_r = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ev_v = [torch.cuda.Event() for _ in range(2)]   # VPP of buffer i done
ev_r = [torch.cuda.Event() for _ in range(2)]   # rSGM has consumed buffer i
pairs = [None, None]

def stage_a(k):
    i = k % 2
    with torch.cuda.stream(s_v):
        s_v.wait_event(ev_r[i])
        o = eng_v.occlusion_heuristic(hints, out=occ[i])
        pairs[i] = eng_v.vpp(left, right, hints, g_occ=o, seed=1)
        ev_v[i].record(s_v)

def stage_b(k):
    i = k % 2
    with torch.cuda.stream(s_r):
        s_r.wait_event(ev_v[i])
        eng_r.rsgm(left, pairs[i][0], pairs[i][1], out=out[i], dmax=D, subpixel=1)
        ev_r[i].record(s_r)

def run(n, pipelined):
    torch.cuda.synchronize(); t = time.perf_counter()
    if pipelined:
        stage_a(0)
        for k in range(n):
            if k + 1 < n: stage_a(k + 1)
            stage_b(k)
    else:
        for k in range(n):
            stage_a(k); stage_b(k)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

for e in ev_r: e.record(s_r)
run(3, False); run(3, True)
print("sequential  %.3f ms/batch" % run(10, False))
print("pipelined   %.3f ms/batch" % run(10, True))
ref = Engine(0)
o = ref.vpp_rsgm(left, right, hints, g_occ=ref.occlusion_heuristic(hints), seed=1, rsgm_kw=dict(dmax=D, subpixel=1))
torch.cuda.synchronize()
print("equal to the fused call:", bool(torch.equal(o, out[1])), bool(torch.equal(o, out[0])))
