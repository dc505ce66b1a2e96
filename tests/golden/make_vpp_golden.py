#!/usr/bin/env python3
"""Generate the VPP golden vectors from the REAL reference implementation.

Runs only in the build container (needs /root/reference, Cython and gcc):
  1. copies nothing into the repo: cythonizes /root/reference/vpp_core/vpp_core_opt.pyx
     into a temporary directory (same recipe as vpp_core/setup.py) and imports it from
     there;
  2. drives ``virtual_projection_scan_rnd`` / ``virtual_projection_scan_max_dist``
     (vpp_core_opt.pyx:53,133) on seeded synthetic inputs;
  3. writes tests/golden/vpp_cases.npz (inputs + expected outputs, small) and
     tests/golden/vpp_anchors.json (SHA-256 of full-size outputs);
  4. writes tests/golden/glibc_rand.json (first outputs of libc rand() per seed).

Only numeric inputs/outputs are stored; no reference source text.
"""
import ctypes
import hashlib
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def build_reference():
    tmp = tempfile.mkdtemp(prefix="vppo_")
    shutil.copy(os.path.join(REF, "vpp_core", "vpp_core_opt.pyx"), tmp)
    shutil.copy(os.path.join(REF, "vpp_core", "setup.py"), tmp)
    subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=tmp,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, tmp)
    import vpp_core_opt  # noqa
    return vpp_core_opt


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def make_inputs(rng, H, W, C, p, dmax_hint, border=False, integer_frac=0.3):
    l = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
    r = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
    g = np.zeros((H, W), np.float32)
    m = rng.random((H, W)) < p
    vals = rng.uniform(0.2, dmax_hint, size=int(m.sum())).astype(np.float32)
    ints = rng.random(vals.shape) < integer_frac
    vals[ints] = np.maximum(1, np.round(vals[ints]))
    half = rng.random(vals.shape) < 0.1
    vals[half] = np.floor(vals[half]) + 0.5  # exercise round-half-away (pyx:82)
    g[m] = vals
    if border:
        # hints hugging all four borders, incl. disparities larger than x (left-side occlusion)
        for (y, x, d) in [(0, 0, 1.0), (0, 1, 1.5), (0, W - 1, 3.25), (H - 1, 0, 0.5), (H - 1, W - 1, 2.0),
                          (H // 2, 0, 2.75), (H // 2, 1, 1.5), (H // 2, 2, 2.5), (1, 3, 3.5), (2, W - 2, 40.0),
                          (H - 2, 2, 2.49), (3, 1, 0.51)]:
            g[y, x] = d
    occ = (rng.random((H, W)) < 0.3).astype(np.uint8)
    return l, r, g, occ


def main():
    v = build_reference()
    libc = ctypes.CDLL("libc.so.6")
    rng = np.random.default_rng(20240229)

    # ------------------------------------------------------------------ glibc rand fixture
    rand_fix = {}
    libc.rand.restype = ctypes.c_int
    for seed in [0, 1, 2, 42, 12345, 2147483647, 2147483648, 4294967295]:
        libc.srand(ctypes.c_uint(seed))
        rand_fix[str(seed)] = [int(libc.rand()) for _ in range(400)]
    with open(os.path.join(HERE, "glibc_rand.json"), "w") as f:
        json.dump(rand_fix, f)

    # ------------------------------------------------------------------ small cases
    inputs = {}
    inputs["rgb"] = make_inputs(rng, 20, 36, 3, 0.08, 14.0)
    inputs["gray"] = make_inputs(rng, 18, 30, 1, 0.10, 12.0)
    inputs["border"] = make_inputs(rng, 12, 24, 3, 0.04, 9.0, border=True)
    inputs["dense"] = make_inputs(rng, 10, 20, 3, 0.5, 6.0)
    store = {}
    for k, (l, r, g, occ) in inputs.items():
        store[f"in_{k}_l"], store[f"in_{k}_r"], store[f"in_{k}_g"], store[f"in_{k}_occ"] = l, r, g, occ

    cases = []
    grid = list(itertools.product(["rnd", "maxdist"], [0, 1], [0, 1], [0, 1], [0, 1], [0.0, 0.3], [1, 3, 5, 7]))
    prng = np.random.default_rng(7)
    prng.shuffle(grid)
    # every value of every factor appears many times in the first 72 draws; maxdist is
    # expensive on the reference so the sample stays small
    picked = grid[:72]
    # make sure the reference defaults are in (test.py:52 rnd, wsize 3, interpolate, c_occ 0)
    picked += [("rnd", 0, 1, 1, 0, 0.0, 3), ("maxdist", 0, 1, 1, 0, 0.0, 3), ("maxdist", 1, 1, 1, 0, 0.0, 3)]
    idx = 0
    for (method, uniform, direction, interp, discard, c_occ, wsize) in picked:
        for inp in (["rgb", "gray", "border", "dense"] if idx % 3 == 0 else [["rgb", "gray", "border", "dense"][idx % 4]]):
            l, r, g, occ = inputs[inp]
            for use_occ in ([1] if idx % 5 else [0, 1]):
                a, b = l.copy(), r.copy()
                H, W, C = a.shape
                o = occ if use_occ else np.zeros_like(occ)
                seed = 1 + (idx % 3)
                v.init_rand(seed)
                if method == "rnd":
                    n = v.virtual_projection_scan_rnd(a, b, g, W, H, C, bool(uniform), wsize, direction, 0.4, c_occ, o,
                                                      bool(discard), bool(interp))
                    agg = (0, 0)
                else:
                    agg = (64, 3) if idx % 2 == 0 else (9, 5)
                    n = v.virtual_projection_scan_max_dist(a, b, g, W, H, C, bool(uniform), wsize, agg[0], agg[1],
                                                           direction, 0.4, c_occ, o, bool(discard), bool(interp))
                name = f"case{len(cases):03d}"
                store[name + "_l"], store[name + "_r"] = a, b
                cases.append(dict(name=name, inp=inp, method=method, uniform=uniform, direction=direction,
                                  interpolate=interp, discard=discard, c=0.4, c_occ=c_occ, wsize=wsize,
                                  agg_x=agg[0], agg_y=agg[1], use_occ=use_occ, seed=seed, n_hints=int(n)))
        idx += 1

    # ------------------------------------------------------------------ maxDistance n_bins==0 fallback (pyx:305-313)
    H, W = 6, 40
    l = np.zeros((H, W, 1), np.uint8)
    r = np.zeros((H, W, 1), np.uint8)
    g = np.zeros((H, W), np.float32)
    g[1, 15] = 0.4          # rows 0..3 x cols 0..31 => 128 L + 128 R zero samples = 256
    g[4, 30] = 2.0
    occ = np.zeros((H, W), np.uint8)
    a, b = l.copy(), r.copy()
    v.init_rand(1)
    n = v.virtual_projection_scan_max_dist(a, b, g, W, H, 1, False, 1, 33, 5, 1, 0.4, 0.0, occ, False, True)
    store["in_fallback_l"], store["in_fallback_r"], store["in_fallback_g"], store["in_fallback_occ"] = l, r, g, occ
    name = f"case{len(cases):03d}"
    store[name + "_l"], store[name + "_r"] = a, b
    assert a.max() > 0, "fallback colour should be 1 -> blend must change an all-zero image"
    cases.append(dict(name=name, inp="fallback", method="maxdist", uniform=0, direction=1, interpolate=1, discard=0,
                      c=0.4, c_occ=0.0, wsize=1, agg_x=33, agg_y=5, use_occ=0, seed=1, n_hints=int(n)))

    np.savez_compressed(os.path.join(HERE, "vpp_cases.npz"), **store)
    with open(os.path.join(HERE, "vpp_cases.json"), "w") as f:
        json.dump(cases, f, indent=0)

    # ------------------------------------------------------------------ full-size anchors (SURVEY App. D recipe)
    H, W, D, p = 540, 960, 192, 0.03
    rng = np.random.default_rng(0)
    l = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    r = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = np.zeros((H, W), np.float32)
    m = rng.random((H, W)) < p
    g[m] = rng.uniform(1, D - 1, size=m.sum()).astype(np.float32)
    occ0 = np.zeros((H, W), np.uint8)
    occ1 = (rng.random((H, W)) < 0.25).astype(np.uint8)
    anchors = dict(recipe="SURVEY.md App. D", H=H, W=W, D=D, p=p,
                   inputs=dict(l=sha(l), r=sha(r), g=sha(g), occ1=sha(occ1)), cases=[])
    for nm, method, occ in [("rnd_occ0", "rnd", occ0), ("rnd_occ1", "rnd", occ1), ("maxdist_occ1", "maxdist", occ1)]:
        a, b = l.copy(), r.copy()
        v.init_rand(1)
        if method == "rnd":
            n = v.virtual_projection_scan_rnd(a, b, g, W, H, 3, False, 3, 1, 0.4, 0.0, occ, False, True)
        else:
            n = v.virtual_projection_scan_max_dist(a, b, g, W, H, 3, False, 3, 64, 3, 1, 0.4, 0.0, occ, False, True)
        anchors["cases"].append(dict(name=nm, n_hints=int(n), l=sha(a), r=sha(b)))
        print(nm, n, sha(a), sha(b))
    # wsize 7 / 5% (TPAMI config, README.md:434-437) and r2l
    a, b = l.copy(), r.copy()
    v.init_rand(3)
    n = v.virtual_projection_scan_rnd(a, b, g, W, H, 3, True, 7, 0, 0.4, 0.25, occ1, False, True)
    anchors["cases"].append(dict(name="rnd_w7_uniform_r2l_cocc", n_hints=int(n), l=sha(a), r=sha(b), seed=3))
    with open(os.path.join(HERE, "vpp_anchors.json"), "w") as f:
        json.dump(anchors, f, indent=1)
    print("cases:", len(cases))


if __name__ == "__main__":
    main()
