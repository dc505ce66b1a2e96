"""Soak test of the fused (lock-step) aggregation layout: many back-to-back steps at the benchmark shape, every result
compared with the first; then the same with unrelated torch kernels running on a second stream (competing for block
slots), which must at worst slow the launches down or trip the bounded polls (reported as an error), never hang."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import synth
from vppstereo_amd.engine import Engine

B, H, W, D = 32, 540, 960, 192
b = synth.make_batch(4, H, W, D, 0.03, seed=1)
dev = torch.device("cuda:0")
rep = lambda a: torch.from_numpy(np.concatenate([a] * (B // 4))).to(dev)
left, right, hints = rep(b["left"]), rep(b["right"]), rep(b["hints"])
eng = Engine()
out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
ref = eng.vpp_rsgm(left, right, hints, seed=1, rsgm_kw=dict(dmax=D, subpixel=1)).clone()
torch.cuda.synchronize()
assert eng.uses_vert() == 3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
bad = 0
for i in range(n):
    eng.vpp_rsgm(left, right, hints, out=out, seed=1, rsgm_kw=dict(dmax=D, subpixel=1))
    if i % 10 == 9:
        bad += int(not torch.equal(out, ref))
torch.cuda.synchronize()
print("alone:", n, "steps,", round((time.perf_counter() - t0) / n * 1e3, 2), "ms per step, mismatching checks:", bad, flush=True)

side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
err = None
t0 = time.perf_counter()
try:
    for i in range(60):
        with torch.cuda.stream(side):
            for _ in range(4):
                a2 = a @ a
        eng.vpp_rsgm(left, right, hints, out=out, seed=1, rsgm_kw=dict(dmax=D, subpixel=1))
        if i % 10 == 9:
            torch.cuda.synchronize()
            bad += int(not torch.equal(out, ref))
    torch.cuda.synchronize()
except Exception as e:  # noqa: BLE001
    err = e
print("with GEMMs on a second stream:", round((time.perf_counter() - t0) / 60 * 1e3, 2), "ms per step, mismatching checks:", bad,
      "layout now", eng.uses_vert(), "error:", err, flush=True)
