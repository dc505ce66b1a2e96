#!/usr/bin/env python3
"""Generate golden vectors for the reference's *Python* glue by importing it.

Runs only in the build container (needs /root/reference).  The reference modules
import numba, cv2 and pyrSGM, none of which is installed here, so they are imported
with stubs: ``numba.njit`` = identity (the functions then run as plain Python),
``cv2``/``pyrSGM`` = empty placeholders (only functions that do not touch them are
driven).  numba typing is emulated where it matters:

  * numba keeps float32 *arrays* but unifies ``n_left = 0`` / float32 loads to float64
    scalars; plain NumPy would compute in float32 (NEP 50).  ``F32Store`` is a float64
    ndarray that rounds to float32 on every store, which reproduces numba's
    "float64 maths, float32 storage" for rsgm._linear_interpolate and
    vpp_standalone._bilateral_filling.
  * uint8 image arithmetic is promoted by numba (SURVEY C-10): images are passed as int64.
  * inputs whose arithmetic is exact in every typing (multiples of 1/8) are used for
    filter.occlusion_heuristic.

Outputs: tests/golden/glue_cases.npz (inputs + expected outputs only).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_stubs():
    numba = types.ModuleType("numba")
    numba.njit = lambda f=None, **kw: f if f is not None else (lambda g: g)
    sys.modules["numba"] = numba
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2GRAY, cv2.COLOR_RGB2GRAY = 6, 7

    def cvt(img, code):
        img = img.astype(np.uint32)
        c0, c1, c2 = img[..., 0], img[..., 1], img[..., 2]
        R, B = (c2, c0) if code == cv2.COLOR_BGR2GRAY else (c0, c2)
        return ((R * 9798 + c1 * 19235 + B * 3735 + 16384) >> 15).astype(np.uint8)

    cv2.cvtColor = cvt
    sys.modules["cv2"] = cv2
    pyr = types.ModuleType("pyrSGM")
    for n in ["census5x5_SSE", "costMeasureCensus5x5_xyd_SSE", "aggregate_SSE", "matchWTA_SSE",
              "matchWTARight_SSE", "subPixelRefine", "median3x3_SSE"]:
        setattr(pyr, n, None)
    sys.modules["pyrSGM"] = pyr
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "models", "rsgm"))


class F32Store(np.ndarray):
    """float64 storage whose stores round to float32 (numba float32-array semantics)."""

    def __setitem__(self, k, v):
        super().__setitem__(k, np.float32(v))


def f32store(a):
    return np.asarray(a, np.float32).astype(np.float64).view(F32Store)


def sparse_disp(rng, h, w, p, lo, hi, step=None):
    d = np.zeros((h, w), np.float32)
    m = rng.random((h, w)) < p
    v = rng.uniform(lo, hi, size=int(m.sum()))
    if step:
        v = np.round(v / step) * step
    d[m] = v.astype(np.float32)
    return d


def main():
    install_stubs()
    import rsgm as ref_rsgm           # models/rsgm/rsgm.py
    import filter as ref_filter       # filter.py
    import vpp_standalone as ref_vpp  # vpp_standalone.py
    import losses as ref_losses       # losses.py

    rng = np.random.default_rng(4242)
    out = {}

    # ---------------------------------------------------------------- _linear_interpolate (rsgm.py:67-113)
    for i, (h, w, p_hole) in enumerate([(12, 40, 0.35), (8, 64, 0.6), (6, 33, 0.2)]):
        d = rng.uniform(1, 60, (h, w)).astype(np.float32)
        # smooth rows so that |nl-nr|<3 often holds
        d = (np.cumsum(rng.normal(0, 0.6, (h, w)), axis=1) + 20 + 3 * rng.random((h, 1))).astype(np.float32)
        d[rng.random((h, w)) < p_hole] = -10.0
        d[rng.random((h, w)) < 0.05] = 0.0
        a = f32store(d)
        ref_rsgm._linear_interpolate(a, 15, 3)
        out[f"linint{i}_in"] = d
        out[f"linint{i}_out"] = np.asarray(a, np.float64).astype(np.float32)

    # ---------------------------------------------------------------- _left_right_check (rsgm.py:230-248)
    for i, (h, w) in enumerate([(10, 48), (7, 31)]):
        dl = rng.uniform(-2, 20, (h, w)).astype(np.float32)
        hm = rng.random((h, w)) < 0.3
        dl[hm] = np.round(dl[hm]) + 0.5  # half-to-even cases
        dr = (dl + rng.normal(0, 0.8, (h, w))).astype(np.float32)
        dr[rng.random((h, w)) < 0.2] = 0
        dl = np.roll(dl, 3, axis=1)
        m = ref_rsgm._left_right_check(dl.copy(), dr.copy(), 1)
        out[f"lrc{i}_dl"], out[f"lrc{i}_dr"], out[f"lrc{i}_mask"] = dl, dr, m

    # ---------------------------------------------------------------- _interpolate_background (rsgm.py:185-227)
    for i, (h, w, p) in enumerate([(12, 40, 0.5), (9, 25, 0.85), (5, 16, 0.0)]):
        d = rng.uniform(1, 90, (h, w)).astype(np.float32)
        d[rng.random((h, w)) < p] = 0
        if i == 1:
            d[:, 7] = 0      # an all-invalid column
            d[4, :] = 0      # an all-invalid row
        a = d.copy()
        ref_rsgm._interpolate_background(a)
        out[f"bg{i}_in"], out[f"bg{i}_out"] = d, a

    # ---------------------------------------------------------------- occlusion_heuristic (filter.py:246-292)
    for i, (h, w, p) in enumerate([(24, 64, 0.06), (16, 48, 0.15)]):
        fg = sparse_disp(rng, h, w, p, 1, 8, step=0.125)
        # a foreground block occluding the background
        fg[h // 3: 2 * h // 3, w // 3: w // 2][fg[h // 3: 2 * h // 3, w // 3: w // 2] > 0] += 9.0
        try:
            dm, conf = ref_filter.occlusion_heuristic(fg.copy())
        except IndexError:
            # SURVEY C-8: interpolate_disparity reads out of bounds under plain Python; the
            # conf_map (the only output test.py:154 uses) is computed before it.
            omap, _ = ref_filter.left_warp(fg.copy())
            cm = ref_filter.weighted_conf(omap, rx=9, ry=7, l=2, g=0.4375, th=1)
            ref_filter.filter(omap, cm, 0.1)
            conf = ref_filter.conf_unwarp(cm, omap)
        out[f"occ{i}_in"], out[f"occ{i}_conf"] = fg, conf

    # ---------------------------------------------------------------- _bilateral_filling (vpp_standalone.py:372-394)
    for i, (h, w, n) in enumerate([(14, 30, 1), (12, 26, 3)]):
        dmap = sparse_disp(rng, h, w, 0.08, 1, 30)
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        img = (img // 64 * 64).astype(np.uint8)  # few grey levels -> non-trivial weights
        aug = ref_vpp._bilateral_filling(f32store(dmap), img.astype(np.int64), n, 2, 1, .001)
        out[f"bil{i}_dmap"], out[f"bil{i}_img"], out[f"bil{i}_n"] = dmap, img, np.int32(n)
        out[f"bil{i}_out"] = np.asarray(aug, np.float64).astype(np.float32)

    # ---------------------------------------------------------------- _get_patch_size_based_on_distance (:7-11)
    # numba computes (d_ref-d_min)/(d_max-d_min) in float32 and the power in float64; plain NumPy
    # would keep float32 throughout, so the float32 ratio is formed here and handed over as
    # (d_ref=ratio, d_min=0, d_max=1) python floats: the reference then does the float64 part.
    ds = np.linspace(1.0, 50.0, 99).astype(np.float32)
    dmin, dmax = np.float32(1.0), np.float32(50.0)
    ns = []
    for d in ds:
        ratio = (d - dmin) / (dmax - dmin)
        assert ratio.dtype == np.float32
        ns.append(ref_vpp._get_patch_size_based_on_distance(float(ratio), 0.0, 1.0, 7, 0.3)[0])
    out["patch_d"], out["patch_n"] = ds, np.asarray(ns, np.int32)

    # ---------------------------------------------------------------- vpp() wrapper structure (:396-432)
    left = rng.integers(0, 256, (9, 14, 3), dtype=np.uint8)
    right = rng.integers(0, 256, (9, 14, 3), dtype=np.uint8)
    lc, rc = ref_vpp.vpp(left, right, np.zeros((9, 14), np.float64))      # early-out :407
    out["vppw_left"], out["vppw_right"], out["vppw_lc0"], out["vppw_rc0"] = left, right, lc, rc
    lcg, rcg = ref_vpp.vpp(left[..., 0], right[..., 0], np.zeros((9, 14), np.float32))  # gray -> [H,W,1] :403
    out["vppw_lcg_shape"] = np.asarray(lcg.shape, np.int32)

    # ---------------------------------------------------------------- losses.guided_metrics (:13-24)
    disp = rng.uniform(0, 50, (11, 17)).astype(np.float32)
    gt = (disp + rng.normal(0, 2, disp.shape)).astype(np.float32)
    valid = (rng.random(disp.shape) < 0.6).astype(np.float32)
    m = ref_losses.guided_metrics(disp.copy(), gt.copy(), valid.copy())
    out["gm_disp"], out["gm_gt"], out["gm_valid"] = disp, gt, valid
    out["gm_out"] = np.asarray([m['bad 1.0'], m['bad 2.0'], m['bad 3.0'], m['bad 4.0'], m['avgerr'], m['rms']],
                               np.float64)

    # ---------------------------------------------------------------- _guided_dsi (rsgm.py:116-127)
    # plain NumPy follows numba here: float32 hint - int64 arange -> float64, everything after it float64,
    # astype(uint16) truncates.  Own generator so that the arrays above keep their values.
    rg = np.random.default_rng(777)
    for i, (h, w, dmax, p) in enumerate([(6, 16, 32, 0.3), (5, 16, 64, 1.0)]):
        dsi = rg.integers(0, 25, (h, w, dmax)).astype(np.uint16)
        valid = (rg.random((h, w)) < p).astype(np.float32)
        hints = (rg.uniform(0, dmax - 1, (h, w)) * valid).astype(np.float32)
        if i == 1:
            hints = np.round(hints * 4) / 4          # quarter-pixel hints: exact zeros of (h - d) included
            hints = hints.astype(np.float32)
        res = ref_rsgm._guided_dsi(dsi.copy(), hints, valid)
        assert res.dtype == np.uint16
        out[f"gdsi{i}_dsi"], out[f"gdsi{i}_hints"], out[f"gdsi{i}_valid"], out[f"gdsi{i}_out"] = dsi, hints, valid, res

    np.savez_compressed(os.path.join(HERE, "glue_cases.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
