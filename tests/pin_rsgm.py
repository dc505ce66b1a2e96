#!/usr/bin/env python3
"""tests/pin_rsgm.py -- one command that pins the rSGM half of the oracle for whoever HAS the reference's natives.

The reference's `pyrSGM` extension (un-vendored submodule thirdparty/stereo-vision, /root/reference/.gitmodules:1-3, imported
at models/rsgm/rsgm.py:6) and OpenCV (rsgm.py:11-12,258-267,285) are absent where this repository was built, so
`oracle/rsgm_oracle.c` restates them from the published algorithm and every rSGM test is "HIP == that restatement".  Given a
real build of either, this script runs INTEGRATION.md section 6's eleven stage checks on seeded inputs:

    python tests/pin_rsgm.py [--pyrsgm-module pyrSGM] [--pyrsgm-path DIR] [--gpu] [--seed N]
    python tests/pin_rsgm.py --dump DIR      # whoever has only THIS repository: inputs.npz + a stand-alone run_reference.py
    python tests/pin_rsgm.py --from DIR      # ... and compares once a maintainer of the reference has sent outputs.npz back

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


H0, W0, D0 = 48, 80, 64


def make_inputs(seed=0):
    """The seeded inputs of the eleven stage checks that do not depend on anybody's outputs."""
    rng = np.random.default_rng(seed)
    H, W, D = H0, W0, D0
    img3 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    # smooth-ish gray pair so that the later stages see structure
    base = (rng.integers(0, 256, (H // 4 + 2, W // 4 + 2)).astype(np.float32))
    base = np.kron(base, np.ones((4, 4), np.float32))[:H, :W + 8]
    gl = np.clip(base[:, 8:] + rng.normal(0, 6, (H, W)), 0, 255).astype(np.uint8)
    gr = np.clip(base[:, :W] + rng.normal(0, 6, (H, W)), 0, 255).astype(np.uint8)
    big = rng.integers(0, 60000, (H, W, D)).astype(np.uint16)   # random costs up to the saturation range (stage 5)
    src = rng.normal(20, 10, (H, W)).astype(np.float32)
    m8 = rng.integers(0, 4, (H, W)).astype(np.uint8) * 40
    m8[10:14, 10:20] = 200  # a 40-pixel island
    return dict(img3=img3, gl=np.ascontiguousarray(gl), gr=np.ascontiguousarray(gr), big=big, src=src, m8=m8)


# The reference side, written so that it needs nothing but numpy and the real modules: `--dump` pastes this function's source
# into the stand-alone script a maintainer of the reference runs (no import of this repository there).
def real_outputs(inp, real, cv2):
    """Outputs of the real natives / cv2 on the seeded inputs, every stage fed by the REAL output of the stage before it.
    real / cv2 may be None: their stages are then absent from the result."""
    import numpy as np
    H, W = inp["gl"].shape
    D = inp["big"].shape[2]
    out = {}
    if cv2 is not None:
        out["gray"] = cv2.cvtColor(inp["img3"], cv2.COLOR_RGB2GRAY)
        out["padded"] = cv2.copyMakeBorder(inp["img3"], 3, 4, 5, 6, cv2.BORDER_REFLECT)
        sp = inp["m8"].copy()
        cv2.filterSpeckles(sp, 0, 200, 10)
        out["speckle"] = sp
    if real is not None:
        cl, cr = np.zeros((H, W), np.uint32), np.zeros((H, W), np.uint32)
        real.census5x5_SSE(inp["gl"], cl, W, H)
        real.census5x5_SSE(inp["gr"], cr, W, H)
        dsi = np.zeros((H, W, D), np.uint16)
        real.costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, W, H, D, 1)
        flat = np.full((H, W), 77, np.uint8)   # flat image: P2 = max(P2min, gamma) everywhere
        agg5 = np.zeros((H, W, D), np.uint16)
        real.aggregate_SSE(flat, inp["big"], agg5, W, H, D, 11, 17, 0.5, 35)
        agg6 = np.zeros((H, W, D), np.uint16)
        real.aggregate_SSE(inp["gl"], dsi, agg6, W, H, D, 7, 12, 0.25, 50)
        # ties and near-ties planted in a copy of the real aggregate: the WTA tie rule and the +-1 exception
        St = agg6.copy()
        St[5, 20, :] = 300; St[5, 20, 7] = 100; St[5, 20, 8] = 100            # exact tie: first minimum wins
        St[6, 30, :] = 300; St[6, 30, 12] = 100; St[6, 30, 13] = 101          # runner-up is the neighbour: valid
        St[7, 40, :] = 300; St[7, 40, 12] = 100; St[7, 40, 30] = 101          # runner-up elsewhere: invalid at uniq 0.95
        St[8, 50, :] = 300; St[8, 50, D - 1] = 50                             # minimum at D-1 (sub-pixel must skip it)
        dl, dr = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
        real.matchWTA_SSE(St, dl, W, H, D, 0.95)
        real.matchWTARight_SSE(St, dr, W, H, D, 0.95)
        sub = dl.copy()
        real.subPixelRefine(St, sub, W, H, D, 0)
        med = np.zeros((H, W), np.float32)
        real.median3x3_SSE(inp["src"], med, W, H)
        out.update(cl=cl, cr=cr, dsi=dsi, agg5=agg5, agg6=agg6, St=St, dl=dl, dr=dr, sub=sub, med=med)
    return out


def compare(inp, out, sides, log=print):
    """Every side on the same inputs (the seeded ones, and the REAL outputs of the stage before where a stage needs them)
    against the real outputs `out`.  Returns (stages run, first failing stage or None)."""
    H, W = inp["gl"].shape
    D = inp["big"].shape[2]
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

    def each(make, call):
        got = {}
        for s, m in sides.items():
            o = make()
            call(m, o)
            got[s] = o
        return got

    if "gray" in out:
        check(1, out["gray"], {s: m.rgb2gray(inp["img3"]) for s, m in sides.items() if hasattr(m, "rgb2gray")})
        check(2, out["padded"], {s: m.pad_reflect(inp["img3"], 3, 4, 5, 6) for s, m in sides.items() if hasattr(m, "pad_reflect")})
    else:
        log("stages 1, 2, 11: no cv2 outputs -> skipped")
    if "cl" in out:
        cl, cr, dsi, St, dl = out["cl"], out["cr"], out["dsi"], out["St"], out["dl"]
        check(3, cl, each(lambda: np.zeros((H, W), np.uint32), lambda m, o: m.census5x5_SSE(inp["gl"], o, W, H)))
        check(4, dsi, each(lambda: np.zeros((H, W, D), np.uint16), lambda m, o: m.costMeasureCensus5x5_xyd_SSE(cl, cr, o, W, H, D, 1)))
        flat = np.full((H, W), 77, np.uint8)
        check(5, out["agg5"], each(lambda: np.zeros((H, W, D), np.uint16), lambda m, o: m.aggregate_SSE(flat, inp["big"], o, W, H, D, 11, 17, 0.5, 35)))
        check(6, out["agg6"], each(lambda: np.zeros((H, W, D), np.uint16), lambda m, o: m.aggregate_SSE(inp["gl"], dsi, o, W, H, D, 7, 12, 0.25, 50)))
        check(7, dl, each(lambda: np.zeros((H, W), np.float32), lambda m, o: m.matchWTA_SSE(St, o, W, H, D, 0.95)))
        check(8, out["dr"], each(lambda: np.zeros((H, W), np.float32), lambda m, o: m.matchWTARight_SSE(St, o, W, H, D, 0.95)))
        check(9, out["sub"], each(lambda: dl.copy(), lambda m, o: m.subPixelRefine(St, o, W, H, D, 0)))
        check(10, out["med"], each(lambda: np.zeros((H, W), np.float32), lambda m, o: m.median3x3_SSE(inp["src"], o, W, H)))
    else:
        log("stages 3-10: no pyrSGM outputs -> skipped")
    if "speckle" in out:
        got = {}
        for s, m in sides.items():
            if hasattr(m, "filterSpeckles"):
                o = inp["m8"].copy()
                m.filterSpeckles(o, 0, 200, 10)   # in place, like cv2's
                got[s] = o
        check(11, out["speckle"], got)
    return ran, failed


def run(real, cv2, sides, seed=0, log=print):
    """real: module with the seven natives or None; cv2: module or None; sides: {name: module with the natives and the
    cv2 restatements}.  Returns (stages run, first failing stage or None)."""
    inp = make_inputs(seed)
    return compare(inp, real_outputs(inp, real, cv2), sides, log)


STANDALONE_HEAD = (
    '#!/usr/bin/env python3\n'
    '"""Stand-alone reference side of vppstereo_amd\'s rSGM pin (written by `tests/pin_rsgm.py --dump`).  Needs ONLY numpy and the\n'
    'reference\'s own natives: `pyrSGM` (bartn8/vppstereo: thirdparty/stereo-vision, built as its README.md:143-146 says; imported at\n'
    'models/rsgm/rsgm.py:6) and/or OpenCV (`cv2`, rsgm.py:11-12,258-267,285).  Run it in this directory:\n\n'
    '    python run_reference.py [--pyrsgm-module pyrSGM] [--pyrsgm-path DIR]\n\n'
    'It reads inputs.npz (seeded arrays), runs the seven natives and the three cv2 calls on them the way rsgm.py does (call sites\n'
    'rsgm.py:25,44,61,141,142,145,170), writes outputs.npz next to it and prints the versions it used.  Send outputs.npz back:\n'
    '`python tests/pin_rsgm.py --from THIS_DIR [--gpu]` compares the CPU oracle and the HIP kernels with it stage by stage."""\n'
    'import importlib\nimport sys\n\nimport numpy as np\n\n\n')
STANDALONE_TAIL = '''

def main():
    name = "pyrSGM"
    if "--pyrsgm-module" in sys.argv:
        name = sys.argv[sys.argv.index("--pyrsgm-module") + 1]
    if "--pyrsgm-path" in sys.argv:
        sys.path.insert(0, sys.argv[sys.argv.index("--pyrsgm-path") + 1])
    real = cv2 = None
    try:
        real = importlib.import_module(name)
    except Exception as e:
        print("no %s module: %s: %s (stages 3-10 skipped)" % (name, type(e).__name__, e))
    try:
        cv2 = importlib.import_module("cv2")
    except Exception as e:
        print("no cv2: %s: %s (stages 1, 2, 11 skipped)" % (type(e).__name__, e))
    inp = dict(np.load("inputs.npz"))
    out = real_outputs(inp, real, cv2)
    out["versions"] = np.array(["numpy " + np.__version__, "cv2 " + (cv2.__version__ if cv2 is not None else "-"),
                                "pyrSGM " + (str(getattr(real, "__file__", "?")) if real is not None else "-")])
    np.savez_compressed("outputs.npz", **out)
    print("wrote outputs.npz:", sorted(k for k in out if k != "versions"))


if __name__ == "__main__":
    main()
'''


def dump(dirname, seed=0):
    """--dump DIR: inputs.npz + the stand-alone run_reference.py (numpy + pyrSGM / cv2 only, no import of this repository)."""
    import inspect
    os.makedirs(dirname, exist_ok=True)
    np.savez_compressed(os.path.join(dirname, "inputs.npz"), seed=np.array(seed), **make_inputs(seed))
    with open(os.path.join(dirname, "run_reference.py"), "w") as f:
        f.write(STANDALONE_HEAD + inspect.getsource(real_outputs) + STANDALONE_TAIL)
    return dirname


def from_dir(dirname, sides, log=print):
    """--from DIR: compare with the outputs.npz a maintainer's run of run_reference.py left in DIR."""
    inp = dict(np.load(os.path.join(dirname, "inputs.npz")))
    inp.pop("seed", None)
    out = dict(np.load(os.path.join(dirname, "outputs.npz")))
    ver = out.pop("versions", None)
    if ver is not None:
        log("[pin_rsgm] reference outputs made with: " + ", ".join(str(v) for v in ver))
    return compare(inp, out, sides, log)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pyrsgm-module", default="pyrSGM", help="name of the real extension module (default pyrSGM)")
    ap.add_argument("--pyrsgm-path", default=None, help="directory to add to sys.path before importing it")
    ap.add_argument("--gpu", action="store_true", help="also compare the HIP drop-in vppstereo_amd.pyrSGM (needs a gfx950 device)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dump", metavar="DIR", default=None, help="write the seeded inputs (inputs.npz) and a stand-alone run_reference.py "
                    "(numpy + pyrSGM / cv2 only) into DIR: what a maintainer of the reference needs to produce outputs.npz")
    ap.add_argument("--from", dest="from_dir", metavar="DIR", default=None, help="compare the oracle (and with --gpu the HIP drop-in) with the "
                    "outputs.npz that run_reference.py left in DIR; needs neither pyrSGM nor cv2 here")
    args = ap.parse_args(argv)
    if args.dump:
        d = dump(args.dump, args.seed)
        print(f"[pin_rsgm] wrote {d}/inputs.npz and {d}/run_reference.py: run `python run_reference.py` there where pyrSGM / cv2 exist, "
              f"then `python tests/pin_rsgm.py --from {d}` here")
        return 0
    if args.from_dir:
        import oracle
        sides = {"oracle (CPU port)": oracle}
        if args.gpu:
            from vppstereo_amd import pyrSGM as hip_natives
            sides["vppstereo_amd.pyrSGM"] = hip_natives
        ran, failed = from_dir(args.from_dir, sides)
        if failed is None:
            print(f"[pin_rsgm] all {ran} stage checks agree with the recorded reference outputs: those stages are PINNED")
            return 0
        print(f"[pin_rsgm] first differing stage: {failed} ({FIX[failed][0]})")
        return 1
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
