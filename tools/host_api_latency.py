#!/usr/bin/env python3
"""tools/host_api_latency.py [out.json]: per-frame latency of the host-pointer drop-ins the reference's test.py calls (numpy in /
numpy out, PCIe inclusive, synchronous, one 540x960x192 frame per call: test.py:154-225 with batch size 1, test.py:293) --
filter.occlusion_heuristic + vpp_standalone.vpp + rsgm.compute_rsgm one after the other, and pipeline.run_frame, the same three
steps in one call (one upload of the pair)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth
from vppstereo_amd import vpp_standalone, rsgm, filter as vfilter, pipeline
H, W, D = 540, 960, 192
fr = synth.make_frame(H, W, D, 0.03, seed=1234)


def T(f, n=30):
    for _ in range(3):
        f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    t.sort()
    return round(t[len(t) // 2] * 1e3, 3)   # median, ms


_, occ = vfilter.occlusion_heuristic(fr["hints"])
lv, rv = vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"], g_occ=occ)
res = {"shape": [H, W, D], "variant": os.environ.get("VPPX_VARIANT"),
       "occlusion_heuristic_ms": T(lambda: vfilter.occlusion_heuristic(fr["hints"])),
       "vpp_ms": T(lambda: vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"], g_occ=occ)),
       "compute_rsgm_ms": T(lambda: rsgm.compute_rsgm(fr["left"], lv, rv, dmax=D)),
       "run_frame_ms": T(lambda: pipeline.run_frame(fr["left"], fr["right"], fr["hints"], maskocc=True, rsgm_kw=dict(dmax=D))),
       "run_frame_with_patterns_ms": T(lambda: pipeline.run_frame(fr["left"], fr["right"], fr["hints"], maskocc=True, rsgm_kw=dict(dmax=D), return_patterns=True)),
       "host_glue_ms": {"np.copy x2": T(lambda: (np.copy(fr["left"]), np.copy(fr["right"]))), "hint_range": T(lambda: vpp_standalone.hint_range(fr["hints"]))}}
res["three_calls_ms"] = round(res["occlusion_heuristic_ms"] + res["vpp_ms"] + res["compute_rsgm_ms"], 3)
res["three_calls_Mdisp_per_s"] = round(H * W * D / res["three_calls_ms"] / 1e3, 1)
res["run_frame_Mdisp_per_s"] = round(H * W * D / res["run_frame_ms"] / 1e3, 1)
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
