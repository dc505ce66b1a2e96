"""tools/clock_trace.py [B]: the shader clock over time while the bench loop runs (tools/build_exp.sh build: VPPX_LIB=tools/bin/libvppx_exp.so).  One extra wave samples
s_memtime (shader cycles) against s_memrealtime (100 MHz) every 20 us while five pipelined steps run."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from vppstereo_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W, D = 540, 960, 192
eng = Engine()
lib = eng.lib
nu = min(B, 8)
b = synth.make_batch(nu, H, W, D, 0.03, seed=1234)
idx = [i % nu for i in range(B)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
outs = [torch.empty((B, H, W), dtype=torch.float32, device=eng.device) for _ in range(2)]
torch.cuda.synchronize()
ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
eng.set_pipeline(True)
def loop(n):
    for k in range(n):
        eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", out=outs[k & 1], rsgm_kw=dict(dmax=D), inputs_ready=ev)
loop(3); torch.cuda.synchronize()
N, PER = 3000, 20
lib.vppx_exp_clock_trace_start.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.vppx_exp_clock_trace_read.argtypes = [C.c_void_p, C.c_void_p]
assert lib.vppx_exp_clock_trace_start(eng.ctx.handle, N, PER) == 0
time.sleep(0.005)
loop(5)
torch.cuda.synchronize()
buf = np.zeros((N, 2), np.uint64)
assert lib.vppx_exp_clock_trace_read(eng.ctx.handle, buf.ctypes.data) == 0
cyc, wall = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
dt = np.diff(wall) / 100.0          # us
mhz = np.diff(cyc) / dt
t = (wall[1:] - wall[0]) / 100.0 / 1000.0   # ms
res = {"B": B, "env": {k: v for k, v in os.environ.items() if k.startswith("VPPX_")}, "mhz_min": float(mhz.min()), "mhz_max": float(mhz.max()),
       "series_ms_mhz": [(round(float(t[i]), 2), int(mhz[i:i + 5].mean())) for i in range(0, len(mhz) - 5, 5)]}
print(json.dumps(res))
