#!/usr/bin/env python3
"""tools/calibrate_cpu.py [out.json]: SURVEY 8d's calibration of the CPU port against the REAL reference, run in the build
container (needs /root/reference, Cython, gcc): the reference's own vpp_core/vpp_core_opt.pyx is cythonized in a temp dir
from where it lies (nothing is copied into the repo) and `virtual_projection_scan_rnd` (pyx:53) is timed on one core
next to `oracle.virtual_projection_scan_rnd` on the same 540x960 inputs, 3 % hints, reference defaults.  The rSGM half of
the port cannot be calibrated: pyrSGM (the reference's SSE natives) is an un-vendored submodule (.gitmodules:1-3)."""
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle  # noqa: E402
import synth   # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_cpu_calibration.json")
    spec = importlib.util.spec_from_file_location("make_vpp_golden", os.path.join(ROOT, "tests", "golden", "make_vpp_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ref = mg.build_reference()
    H, W, D, P = 540, 960, 192, 0.03
    rows = []
    for f in range(4):
        fr = synth.make_frame(H, W, D, P, seed=1234, frame=f)
        occ = np.zeros((H, W), np.uint8)
        t_ref, t_port = [], []
        for rep in range(5):
            l, r = fr["left"].copy(), fr["right"].copy()
            ref.init_rand(1 + f)
            t0 = time.perf_counter()
            n_ref = ref.virtual_projection_scan_rnd(l, r, fr["hints"], W, H, 3, False, 3, 1, 0.4, 0.0, occ, False, True)
            t_ref.append(time.perf_counter() - t0)
            l2, r2 = fr["left"].copy(), fr["right"].copy()
            oracle.init_rand(1 + f)
            t0 = time.perf_counter()
            n_port = oracle.virtual_projection_scan_rnd(l2, r2, fr["hints"], W, H, 3, False, 3, 1, 0.4, 0.0, occ, False, True)
            t_port.append(time.perf_counter() - t0)
            assert n_ref == n_port and np.array_equal(l, l2) and np.array_equal(r, r2), "port != reference"
        rows.append({"frame": f, "hints": int(n_ref), "reference_cython_ms": round(min(t_ref) * 1e3, 2), "oracle_port_ms": round(min(t_port) * 1e3, 2)})
    a = sum(r["reference_cython_ms"] for r in rows) / len(rows)
    b = sum(r["oracle_port_ms"] for r in rows) / len(rows)
    doc = {"what": "SURVEY 8d calibration: the reference's Cython virtual_projection_scan_rnd (vpp_core_opt.pyx:53, cythonized from /root/reference, "
                   "gcc -O2 as its setup.py builds it) against the CPU port oracle/vpp_oracle.c (gcc " + open(os.path.join(ROOT, "oracle", "Makefile")).read().split("CFLAGS ?=")[1].split("\n")[0].strip() +
                   "), one core of the build container, 540x960 RGB, 3 % hints, best of 5, outputs bit-identical",
           "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else None,
           "frames": rows, "reference_cython_ms_mean": round(a, 2), "oracle_port_ms_mean": round(b, 2), "port_over_reference_time": round(b / a, 3),
           "rsgm": "cannot be calibrated: pyrSGM (census5x5_SSE ... median3x3_SSE, rsgm.py:6) is an un-vendored submodule (.gitmodules:1-3); the port's "
                   "rSGM is scalar C, and rSGM is > 99 % of the port's 4.9 s per frame, so cpu_baseline is a lower bound of what the reference's SSE natives "
                   "would do on one core"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
