"""tools/overlap_probe.py [B H W D]: sum / WTA kernel and a W/E launch of the same part, alone and started together on two
streams (vppx_exp_overlap of a `tools/build_exp.sh` build: VPPX_LIB=tools/bin/libvppx_exp.so).  The question: do the two
big kernels with complementary stalls (memory-wait vs issue-wait) overlap, i.e. is pair < sum + we?"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine

B, H, W, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 540, 960, 192)
eng = Engine()
nu = min(B, 8)
b = synth.make_batch(nu, H, W, D, 0.03, seed=1234)
idx = [i % nu for i in range(B)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
out = eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", rsgm_kw=dict(dmax=D))
torch.cuda.synchronize()
assert eng.uses_vert() == 3
f = eng.lib.vppx_exp_overlap
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
res = {"shape": (B, H, W, D)}
ms = (C.c_float * 3)()
for rep in range(2):
    for mode, name in ((1, "sum_alone"), (2, "we_alone"), (3, "pair")):
        rc = f(eng.ctx.handle, mode, 8, ms)
        assert rc == 0, eng.lib.vppx_last_error()
        res[f"{name}_{rep}"] = {"wall": round(ms[0], 4), "sum": round(ms[1], 4), "we": round(ms[2], 4)}
print(json.dumps(res), flush=True)
