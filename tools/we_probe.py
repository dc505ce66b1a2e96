"""tools/we_probe.py [B [steps]]: the bench loop (occlusion heuristic + VPP + rSGM, inputs resident, cross-step overlap on
and off) with the in-step duration of BOTH aggregation launches (event pairs on the launch stream) next to their
back-to-back re-launch figures.  The environment's VPPX_* knobs (experiment hooks need a `tools/build_exp.sh` build, VPPX_LIB=tools/bin/libvppx_exp.so) are echoed."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, W, D = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (540, 960, 192)
eng = Engine()
nu = min(B, 8)
b = synth.make_batch(nu, H, W, D, 0.03, seed=1234)
idx = [i % nu for i in range(B)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
outs = [torch.empty((B, H, W), dtype=torch.float32, device=eng.device) for _ in range(2)]
occ = torch.empty((B, H, W), dtype=torch.uint8, device=eng.device)
torch.cuda.synchronize()
ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()


def loop(n):
    for k in range(n):
        eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", occ_out=occ, out=outs[k & 1], rsgm_kw=dict(dmax=D), inputs_ready=ev)


res = {"shape": (B, H, W, D), "env": {k: v for k, v in os.environ.items() if k.startswith("VPPX_")}}
for pipe in (True, False):
    eng.set_pipeline(pipe)
    loop(3)
    torch.cuda.synchronize()
    eng.agg_kernel_ms(0); eng.we_kernel_ms(0)
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    tag = "piped" if pipe else "unpiped"
    res[tag] = {"step_ms": round(ms, 3), "we_in_step": round(eng.we_kernel_ms(steps)[0], 3), "vert_in_step": round(eng.agg_kernel_ms(steps)[0], 3)}
res["layout"] = eng.uses_vert()
if eng.uses_vert() == 3:
    res["we_b2b"] = round(eng.time_aggregate_part(1, 5), 3)
    res["vert_b2b"] = round(eng.time_aggregate_part(2, 5), 3)
try:
    eng.status()
except Exception as e:
    res["status"] = str(e)[:80]
print(json.dumps(res), flush=True)
