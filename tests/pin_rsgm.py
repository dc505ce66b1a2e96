#!/usr/bin/env python3
"""tests/pin_rsgm.py -- one command that pins the rSGM half of the oracle for whoever HAS the reference's natives.

The reference's `pyrSGM` extension (un-vendored submodule thirdparty/stereo-vision, /root/reference/.gitmodules:1-3, imported
at models/rsgm/rsgm.py:6) and OpenCV (rsgm.py:11-12,258-267,285) are absent where this repository was built, so
`oracle/rsgm_oracle.c` restates them from the published algorithm and every rSGM test is "HIP == that restatement".  Given a
real build of either, this script runs INTEGRATION.md section 6's eleven stage checks on seeded inputs:

    python tests/pin_rsgm.py [--pyrsgm-module pyrSGM] [--pyrsgm-path DIR] [--gpu] [--seed N]

Every stage is fed the SAME inputs on all sides (stages are independent: the real module's output of stage k is the input
of stage k+1 everywhere), so the first FAIL names the stage that differs, and the table row says which constant to flip
(`oracle/rsgm_oracle.c` and the kernel named there).  Sides: the real module, the CPU oracle and -- with --gpu on a gfx950
box -- the HIP drop-in `vppstereo_amd.pyrSGM`.  Without pyrSGM and without cv2 it prints why it has nothing to compare with
and exits 0.  Exit status: 0 = every stage run agrees (or nothing to run), 1 = a stage differs.

This is test infrastructure (it loads the oracle): it lives under tests/, `tests/test_pin_rsgm.py` drives it with stand-in
modules (the oracle itself, and a deliberately altered copy whose first differing stage must be found).
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# INTEGRATION.md section 6, one row per stage: (number, name, what to change when it differs)
FIX = {
    1: ("gray conversion", "gray weights 9798/19235/3735 >> 15 (OpenCV >= 3.4) vs 4899/9617/1868 >> 14: `pad_gray_kernel`, "
                           "`pad_gray_census_kernel`, `gray_ctx_kernel` / `rsgmo_rgb2gray`"),
    2: ("reflect padding", "`reflect_idx` / `rsgmo_pad_reflect` (BORDER_REFLECT, lo = pad // 2)"),
    3: ("census 5x5", "comparison sense, bit order or the 2-pixel border: `census5x5_kernel`, `census_of` / `rsgmo_census5x5`"),
    4: ("Hamming cost volume", "`INVALID_DISP_COST` (cost of d > x; 16 here) on both sides: `cost_kernel`, `step_costs` / "
                               "`rsgmo_cost_census5x5_xyd`"),
    5: ("8-path aggregation (flat image: P2 constant)", "saturation points / the d+-1 border value / the first pixel of a path: "
                                                        "`sgm_update`, `sgm_update_split` / `rsgmo_aggregate_paths`"),
    6: ("adaptive P2 (textured image)", "`p2_lut_host` / `rsgmo_p2_lut` (float32, no contraction) and the image the binding "
                                        "reads: `vppx_aggregate_img`"),
    7: ("left WTA", "tie rule, the best+-1 exception, the invalid marker -10: `wta_rows`, `top2_final` / `rsgmo_match_wta`"),
    8: ("right WTA", "search range at the right border: `sum_wta_lr_kernel` right view, `wta_right_kernel` / `rsgmo_match_wta_right`"),
    9: ("sub-pixel refinement", "range conditions (1 <= x <= W-2, disp > 0, 1 <= d <= D-2) and the float32 expression: "
                                "`wta_rows` sub-pixel block / `rsgmo_subpixel_refine`"),
    10: ("median 3x3", "border columns (copied here): `median3x3_kernel`, `median_interp_clip_kernel` / `rsgmo_median3x3`"),
    11: ("filterSpeckles", "connectivity / the newVal exclusion / size threshold: `speckle_*_kernel` / `rsgmo_filter_speckles_u8`"),
}


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and bool(np.array_equal(a, b))


def _where(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return f"shapes {a.shape} vs {b.shape}"
    bad = np.argwhere(a != b)
    i = tuple(int(v) for v in bad[0])
    return f"{len(bad)} of {a.size} elements differ, first at {i}: real {a[i]} vs ours {b[i]}"


def run(real, cv2, sides, seed=0, log=print):
    """real: module with the seven natives or None; cv2: module or None; sides: {name: module with the natives and the
    cv2 restatements}.  Returns (stages run, first failing stage or None)."""
    rng = np.random.default_rng(seed)
    H, W, D = 48, 80, 64
    failed, ran = None, 0

    def check(stage, want, got_by_side):
        nonlocal failed, ran
        ran += 1
        name, fix = FIX[stage]
        for side, got in got_by_side.items():
            if _same(want, got):
                log(f"stage {stage:2d} {name:45s} {side:22s} PASS")
            else:
                log(f"stage {stage:2d} {name:45s} {side:22s} FAIL  {_where(want, got)}")
                if failed is None:
                    failed = stage
                    log(f"    -> first differing stage.  To change: {fix}")

    img3 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    # smooth-ish gray pair so that the later stages see structure
    base = (rng.integers(0, 256, (H // 4 + 2, W // 4 + 2)).astype(np.float32))
    base = np.kron(base, np.ones((4, 4), np.float32))[:H, :W + 8]
    gl = np.clip(base[:, 8:] + rng.normal(0, 6, (H, W)), 0, 255).astype(np.uint8)
    gr = np.clip(base[:, :W] + rng.normal(0, 6, (H, W)), 0, 255).astype(np.uint8)

    if cv2 is not None:
        want = cv2.cvtColor(img3, cv2.COLOR_RGB2GRAY)
        check(1, want, {s: m.rgb2gray(img3) for s, m in sides.items() if hasattr(m, "rgb2gray")})
        want = cv2.copyMakeBorder(img3, 3, 4, 5, 6, cv2.BORDER_REFLECT)
        check(2, want, {s: m.pad_reflect(img3, 3, 4, 5, 6) for s, m in sides.items() if hasattr(m, "pad_reflect")})
    else:
        log("stages 1, 2, 11: cv2 is not importable here -> skipped")

    if real is not None:
        def native(mod, fn, *args):
            getattr(mod, fn)(*args)

        # 3: census
        cl, cr = np.zeros((H, W), np.uint32), np.zeros((H, W), np.uint32)
        native(real, "census5x5_SSE", gl, cl, W, H)
        native(real, "census5x5_SSE", gr, cr, W, H)
        got = {}
        for s, m in sides.items():
            o = np.zeros((H, W), np.uint32)
            m.census5x5_SSE(gl, o, W, H)
            got[s] = o
        check(3, cl, got)
        # 4: cost volume from the REAL census images
        dsi = np.zeros((H, W, D), np.uint16)
        native(real, "costMeasureCensus5x5_xyd_SSE", cl, cr, dsi, W, H, D, 1)
        got = {}
        for s, m in sides.items():
            o = np.zeros((H, W, D), np.uint16)
            m.costMeasureCensus5x5_xyd_SSE(cl, cr, o, W, H, D, 1)
            got[s] = o
        check(4, dsi, got)
        # 5: aggregation with a flat image (P2 = max(P2min, gamma) everywhere), random costs up to the saturation range
        big = rng.integers(0, 60000, (H, W, D)).astype(np.uint16)
        flat = np.full((H, W), 77, np.uint8)
        for stage, image, costs, (p1, p2min, alpha, gamma) in ((5, flat, big, (11, 17, 0.5, 35)), (6, gl, dsi, (7, 12, 0.25, 50))):
            agg = np.zeros((H, W, D), np.uint16)
            native(real, "aggregate_SSE", image, costs, agg, W, H, D, p1, p2min, alpha, gamma)
            got = {}
            for s, m in sides.items():
                o = np.zeros((H, W, D), np.uint16)
                m.aggregate_SSE(image, costs, o, W, H, D, p1, p2min, alpha, gamma)
                got[s] = o
            check(stage, agg, got)
        S = agg  # the real module's aggregate of the real costs (stage 6's inputs)
        # plant ties and near-ties in a copy: the WTA tie rule and the +-1 exception
        St = S.copy()
        St[5, 20, :] = 300; St[5, 20, 7] = 100; St[5, 20, 8] = 100            # exact tie: first minimum wins
        St[6, 30, :] = 300; St[6, 30, 12] = 100; St[6, 30, 13] = 101          # runner-up is the neighbour: valid
        St[7, 40, :] = 300; St[7, 40, 12] = 100; St[7, 40, 30] = 101          # runner-up elsewhere: invalid at uniq 0.95
        St[8, 50, :] = 300; St[8, 50, D - 1] = 50                             # minimum at D-1 (sub-pixel must skip it)
        for stage, fn in ((7, "matchWTA_SSE"), (8, "matchWTARight_SSE")):
            want = np.zeros((H, W), np.float32)
            native(real, fn, St, want, W, H, D, 0.95)
            got = {}
            for s, m in sides.items():
                o = np.zeros((H, W), np.float32)
                getattr(m, fn)(St, o, W, H, D, 0.95)
                got[s] = o
            check(stage, want, got)
            if stage == 7:
                dl = want
        # 9: sub-pixel on the REAL left disparities
        want = dl.copy()
        native(real, "subPixelRefine", St, want, W, H, D, 0)
        got = {}
        for s, m in sides.items():
            o = dl.copy()
            m.subPixelRefine(St, o, W, H, D, 0)
            got[s] = o
        check(9, want, got)
        # 10: median
        src = rng.normal(20, 10, (H, W)).astype(np.float32)
        want = np.zeros((H, W), np.float32)
        native(real, "median3x3_SSE", src, want, W, H)
        got = {}
        for s, m in sides.items():
            o = np.zeros((H, W), np.float32)
            m.median3x3_SSE(src, o, W, H)
            got[s] = o
        check(10, want, got)
    else:
        log("stages 3-10: no pyrSGM module importable here -> skipped")

    if cv2 is not None:
        m8 = rng.integers(0, 4, (H, W)).astype(np.uint8) * 40
        m8[10:14, 10:20] = 200  # a 40-pixel island
        want = m8.copy()
        cv2.filterSpeckles(want, 0, 200, 10)
        got = {}
        for s, m in sides.items():
            if hasattr(m, "filterSpeckles"):
                o = m8.copy()
                m.filterSpeckles(o, 0, 200, 10)   # in place, like cv2's
                got[s] = o
        check(11, want, got)
    return ran, failed


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pyrsgm-module", default="pyrSGM", help="name of the real extension module (default pyrSGM)")
    ap.add_argument("--pyrsgm-path", default=None, help="directory to add to sys.path before importing it")
    ap.add_argument("--gpu", action="store_true", help="also compare the HIP drop-in vppstereo_amd.pyrSGM (needs a gfx950 device)")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    if args.pyrsgm_path:
        sys.path.insert(0, args.pyrsgm_path)
    real = cv2 = None
    try:
        real = importlib.import_module(args.pyrsgm_module)
    except Exception as e:  # noqa: BLE001
        print(f"[pin_rsgm] no `{args.pyrsgm_module}` module here ({type(e).__name__}: {e})")
    try:
        cv2 = importlib.import_module("cv2")
    except Exception as e:  # noqa: BLE001
        print(f"[pin_rsgm] no cv2 here ({type(e).__name__}: {e})")
    if real is None and cv2 is None:
        print("[pin_rsgm] skipped: nothing to compare with.  Build the reference's thirdparty/stereo-vision (pyrSGM) and/or install "
              "OpenCV, then run this again; rSGM parity stays UNPINNED until then (DESIGN.md section 2).")
        return 0
    import oracle
    sides = {"oracle (CPU port)": oracle}
    if args.gpu:
        from vppstereo_amd import pyrSGM as hip_natives
        sides["vppstereo_amd.pyrSGM"] = hip_natives
    ran, failed = run(real, cv2, sides, seed=args.seed)
    if failed is None:
        print(f"[pin_rsgm] all {ran} stage checks agree: the stages run are PINNED against this build of "
              f"{'pyrSGM' if real is not None else ''}{' and ' if real is not None and cv2 is not None else ''}{'cv2 ' + cv2.__version__ if cv2 is not None else ''}")
        return 0
    print(f"[pin_rsgm] first differing stage: {failed} ({FIX[failed][0]})")
    return 1


if __name__ == "__main__":
    sys.exit(main())
