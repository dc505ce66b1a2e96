"""tools/soak_shapes.py [steps]: the pipelined loop with the occlusion mask at shapes that take the round-5 / round-6 code paths -- 32 and 12 frames of
540x960x192 (W/E with its last layer of waves cut into pieces), 20 frames of
540x960x192 (a whole lock-step round + 4: two lock-step launches, W/E next to the second), 16 frames of 375x1242x192 (12 + 4),
8 frames of 540x960x192 (W/E next to an under-filled launch), 8 frames of 1536x2048x256 (trapezoid ring) -- for `steps` back-to-back
steps each, every tenth result compared with the first and with the 8-path layout; prints ms per step and mismatches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
os.environ["VPPX_VERT"] = "0"
ref_eng = Engine()
os.environ.pop("VPPX_VERT")
eng = Engine()
eng.set_pipeline(True)
dev = eng.device
total_bad = 0
# (round 6: 32 and 12 frames of 540x960x192 run their W/E launches with the last layer of waves cut into 4 / 5 dependent pieces)
for (B, H, W, D, p) in ((32, 540, 960, 192, 0.03), (12, 540, 960, 192, 0.03), (20, 540, 960, 192, 0.03), (16, 375, 1242, 192, 0.05), (8, 540, 960, 192, 0.03),
                        (8, 1536, 2048, 256, 0.01)):
    nu = min(B, 4)
    b = synth.make_batch(nu, H, W, D, p, seed=B + H)
    idx = [i % nu for i in range(B)]
    l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(dev) for k in ("left", "right", "hints"))
    want = ref_eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", seed=1, rsgm_kw=dict(dmax=D, subpixel=1)).clone()
    ref_eng.synchronize()
    outs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
    bad = 0
    t0 = time.perf_counter()
    for i in range(n):
        eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", out=outs[i & 1], seed=1, rsgm_kw=dict(dmax=D, subpixel=1), inputs_ready=ev)
        if i % 10 == 9:
            bad += int(not torch.equal(outs[i & 1], want))
    eng.synchronize()
    total_bad += bad
    print(f"{B} x {H}x{W}x{D}: {n} steps, {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step (incl. the checks), layout {eng.uses_vert()}, "
          f"parts {eng.last_call_parts()}, mismatching checks {bad}, lost lock steps {eng.ctx.lockstep_failures}", flush=True)
print("SOAK_OK" if total_bad == 0 and eng.ctx.lockstep_failures == 0 else "SOAK_FAILED", flush=True)
