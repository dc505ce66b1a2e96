#!/usr/bin/env python3
"""Per-frame latency of the host-pointer drop-ins (numpy in / numpy out, PCIe inclusive)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth
from vppstereo_amd import vpp_standalone, rsgm, filter as vfilter
H, W, D = 540, 960, 192
fr = synth.make_frame(H, W, D, 0.03, seed=1234)
for _ in range(3):
    lv, rv = vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"])
    out = rsgm.compute_rsgm(fr["left"], lv, rv, dmax=D)
n = 20
t0 = time.perf_counter()
for _ in range(n):
    _, occ = vfilter.occlusion_heuristic(fr["hints"])
t_occ = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    lv, rv = vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"], g_occ=occ)
t_vpp = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    out = rsgm.compute_rsgm(fr["left"], lv, rv, dmax=D)
t_rsgm = (time.perf_counter() - t0) / n
tot = t_occ + t_vpp + t_rsgm
print(f"host API per 540x960x192 frame: occlusion_heuristic {t_occ*1e3:.2f} ms, vpp {t_vpp*1e3:.2f} ms, compute_rsgm {t_rsgm*1e3:.2f} ms "
      f"-> {H*W*D/tot/1e6:.0f} Mdisp/s PCIe-inclusive, single frame, synchronous")
