"""Cross-call pipelining (Engine.set_pipeline): a stream of DIFFERENT batches, results against the unpipelined engine,
then the step time of both modes at the benchmark shape."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import synth
from vppstereo_amd.engine import Engine

dev = torch.device("cuda:0")
B, H, W, D = 32, 540, 960, 192
base = [synth.make_batch(4, H, W, D, 0.03, seed=s) for s in (1, 2, 3)]
def rep(a): return torch.from_numpy(np.concatenate([a] * (B // 4))).to(dev)
batches = [tuple(rep(b[k]) for k in ("left", "right", "hints")) for b in base]
ref_eng, eng = Engine(), Engine()
eng.set_pipeline(True)

def run(e, n, outs, occs):
    for i in range(n):
        l, r, h = batches[i % 3]
        occ = e.occlusion_heuristic(h, out=occs[i % 2])
        e.vpp_rsgm(l, r, h, g_occ=occ, out=outs[i % len(outs)], seed=i % 3 + 1, rsgm_kw=dict(dmax=D, subpixel=1))

occs = [torch.empty((B, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
refs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(3)]
run(ref_eng, 3, refs, occs)
torch.cuda.synchronize()
outs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(6)]
run(eng, 6, outs, occs)
torch.cuda.synchronize()
bad = [i for i in range(6) if not torch.equal(outs[i], refs[i % 3])]
print("pipelined stream of 6 batches (3 different): mismatching batches", bad, "layout", eng.uses_vert(), flush=True)
for e, name in ((ref_eng, "unpipelined"), (eng, "pipelined  "), (ref_eng, "unpipelined"), (eng, "pipelined  ")):
    run(e, 4, outs, occs); torch.cuda.synchronize()
    n = 12
    t0 = time.perf_counter(); run(e, n, outs, occs); torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / n * 1e3, 3), "ms per step", flush=True)
run(eng, 6, outs, occs); torch.cuda.synchronize()
print("after timing loops: mismatching batches", [i for i in range(6) if not torch.equal(outs[i], refs[i % 3])])
