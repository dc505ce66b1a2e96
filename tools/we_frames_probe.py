"""tools/we_frames_probe.py [H W D]: the W/E launch (`sgm_we12_kernel`) re-launched back to back over 12 .. 18 frames IN ONE PROCESS (a process
keeps whatever makes its W/E launches fast or slow -- round 6 found them bimodal between processes, 0.81 / 0.88 ms per 16 frames -- so the
frame counts are compared inside it), twice over to show the repeatability.  ms per launch, per frame, and waves per SIMD."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine

H, W, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (540, 960, 192)
eng = Engine()
b = synth.make_batch(8, H, W, D, 0.03, seed=1234)
idx = [i % 8 for i in range(18)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
Hp = -(-H // 16) * 16
rows = []
for rep in range(2):
    for B in (16, 15, 17, 14, 12, 18, 16):
        out = eng.vpp_rsgm(l[:B], r[:B], h[:B], g_occ="occlusion_heuristic", rsgm_kw=dict(dmax=D))
        torch.cuda.synchronize()
        if eng.uses_vert() != 3 or eng.last_call_parts() != 1:
            rows.append({"frames": B, "skipped": f"layout {eng.uses_vert()}, parts {eng.last_call_parts()}"})
            continue
        ms = eng.time_aggregate_part(1, 10)
        rows.append({"rep": rep, "frames": B, "we_ms": round(ms, 4), "ms_per_frame": round(ms / B, 5),
                     "waves_per_simd": round(B * 2 * Hp / 4 / 1024, 3)})
print(json.dumps({"shape": [H, W, D], "rows": rows}))
