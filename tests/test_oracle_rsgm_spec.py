"""The rSGM natives are not in the reference's tree (parity unpinned), so the C oracle is
checked against known answers derived from the written spec: hand-computed
tiny cases and an independent pure-Python restatement of each stage."""
import numpy as np
import pytest

import oracle


def py_census(img):
    h, w = img.shape
    out = np.zeros((h, w), np.uint32)
    for y in range(2, h - 2):
        for x in range(2, w - 2):
            v = 0
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    if dy == 0 and dx == 0:
                        continue
                    v = (v << 1) | int(img[y + dy, x + dx] < img[y, x])
            out[y, x] = v
    return out


def py_cost(cl, cr, D):
    h, w = cl.shape
    c = np.full((h, w, D), 16, np.uint16)
    for y in range(h):
        for x in range(w):
            for d in range(min(D - 1, x) + 1):
                c[y, x, d] = bin(int(cl[y, x]) ^ int(cr[y, x - d])).count("1")
    return c


def py_aggregate(img, dsi, p1, p2min, alpha, gamma, paths=range(8)):
    DX = [-1, -1, 0, 1, 1, 1, 0, -1]
    DY = [0, -1, -1, -1, 0, 1, 1, 1]
    h, w, D = dsi.shape
    sat = lambda v: min(v, 65535)
    S = np.zeros((h, w, D), np.int64)
    for k in paths:
        L = np.zeros((h, w, D), np.int64)
        fwd = k < 4
        ys = range(h) if fwd else range(h - 1, -1, -1)
        xs = range(w) if fwd else range(w - 1, -1, -1)
        for y in ys:
            for x in xs:
                py, px = y + DY[k], x + DX[k]
                C = dsi[y, x].astype(np.int64)
                if not (0 <= py < h and 0 <= px < w):
                    L[y, x] = C
                else:
                    Lp = L[py, px]
                    di = abs(int(img[y, x]) - int(img[py, px]))
                    p2 = max(p2min, int(np.float32(np.float32(-alpha) * np.float32(di)) + np.float32(gamma)))
                    mn = int(Lp.min())
                    for d in range(D):
                        m = int(Lp[d])
                        if d > 0:
                            m = min(m, sat(int(Lp[d - 1]) + p1))
                        if d < D - 1:
                            m = min(m, sat(int(Lp[d + 1]) + p1))
                        m = min(m, sat(mn + p2))
                        L[y, x, d] = sat(int(C[d]) + m) - mn
        S = np.minimum(S + L, 65535)
    return S.astype(np.uint16)


def py_wta(c, uniq):
    f = int(np.float32(1024.0) * np.float32(uniq))
    best = int(np.argmin(c))
    minc = int(c[best])
    others = [int(v) for i, v in enumerate(c) if i != best]
    sec = min(others) if others else 65535
    if 1024 * minc <= sec * f:
        return float(best)
    if best > 0 and int(c[best - 1]) == sec:
        return float(best)
    if best + 1 < len(c) and int(c[best + 1]) == sec:
        return float(best)
    return -10.0


def test_census_bit_order_and_border():
    img = np.full((5, 5), 9, np.uint8)
    out = np.zeros((5, 5), np.uint32)
    oracle.census5x5_SSE(img, out, 5, 5)
    assert not out.any()                      # nothing is smaller than the centre
    img[0, 0] = 1                             # first neighbour in row-major order -> bit 23
    oracle.census5x5_SSE(img, out, 5, 5)
    assert out[2, 2] == 1 << 23 and np.count_nonzero(out) == 1
    img[0, 0], img[4, 4] = 9, 1               # last neighbour -> bit 0
    oracle.census5x5_SSE(img, out, 5, 5)
    assert out[2, 2] == 1
    img[4, 4], img[2, 3] = 9, 1               # right neighbour of the centre: index 12 of 24 -> bit 11
    oracle.census5x5_SSE(img, out, 5, 5)
    assert out[2, 2] == 1 << 11
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (9, 16), dtype=np.uint8)
    got = np.zeros(big.shape, np.uint32)
    oracle.census5x5_SSE(big, got, 16, 9)
    assert np.array_equal(got, py_census(big))
    assert not got[:2].any() and not got[-2:].any() and not got[:, :2].any() and not got[:, -2:].any()


def test_cost_popcount_and_invalid_region():
    rng = np.random.default_rng(1)
    cl = rng.integers(0, 1 << 24, (3, 16), dtype=np.uint32)
    cr = rng.integers(0, 1 << 24, (3, 16), dtype=np.uint32)
    dsi = np.zeros((3, 16, 8), np.uint16)
    oracle.costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, 16, 3, 8)
    assert np.array_equal(dsi, py_cost(cl, cr, 8))
    assert (dsi[:, 0, 1:] == 16).all() and dsi[1, 5, 5] == bin(int(cl[1, 5]) ^ int(cr[1, 0])).count("1")


def test_aggregate_hand_computed_two_pixels():
    """One row, two pixels, D = 3, path W only: L(0) = C(0); L(1,d) = C + min(Lp(d), Lp(d+-1)+P1, min+P2) - min."""
    img = np.array([[10, 14] + [14] * 14], np.uint8)
    dsi = np.zeros((1, 16, 8), np.uint16)
    dsi[0, 0, :3] = [5, 1, 9]
    dsi[0, 1, :3] = [2, 7, 0]
    dsi[:, :, 3:] = 100
    S = np.zeros_like(dsi)
    oracle.aggregate_SSE(img, dsi, S, 16, 1, 8, 3, 4, 0.5, 10, path_mask=1)      # P1 = 3, P2 = max(4, int(-0.5*4+10)) = 8
    assert S[0, 0, :3].tolist() == [5, 1, 9]
    # Lp = [5,1,9,100..], min = 1: d0: min(5, 1+3, 1+8)=4 -> 2+4-1=5 ; d1: min(1, 5+3, 9+3, 9)=1 -> 7+1-1=7 ; d2: min(9,1+3,100+3,9)=4 -> 0+4-1=3
    assert S[0, 1, :3].tolist() == [5, 7, 3]


@pytest.mark.parametrize("params", [(11, 17, 0.5, 35), (3, 4, 2.0, 9), (70, 20, 0.25, 900)])
def test_aggregate_vs_python_restatement(params):
    rng = np.random.default_rng(7)
    h, w, D = 6, 16, 8
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    dsi = rng.integers(0, 40, (h, w, D)).astype(np.uint16)
    p1, p2min, alpha, gamma = params
    for mask in (0xFF, 1 << 1, 1 << 3, 1 << 5, 1 << 6):
        S = np.zeros_like(dsi)
        oracle.aggregate_SSE(img, dsi, S, w, h, D, p1, p2min, alpha, gamma, path_mask=mask)
        want = py_aggregate(img, dsi, p1, p2min, alpha, gamma, [k for k in range(8) if (mask >> k) & 1])
        assert np.array_equal(S, want), (params, mask)


def test_wta_rules_and_right_view():
    rng = np.random.default_rng(3)
    h, w, D = 4, 16, 8
    S = rng.integers(0, 50, (h, w, D)).astype(np.uint16)
    S[0, 9] = [30, 30, 5, 30, 30, 30, 30, 30]      # clear winner
    S[0, 10] = [30, 30, 20, 20, 30, 30, 30, 30]    # tie: first minimum, runner-up adjacent -> valid
    S[0, 11] = [30, 20, 30, 30, 30, 20, 30, 30]    # ambiguous, runner-up far -> invalid
    dl = np.zeros((h, w), np.float32)
    oracle.matchWTA_SSE(S, dl, w, h, D, 0.95)
    assert dl[0, 9] == 2 and dl[0, 10] == 2 and dl[0, 11] == -10 and dl[0, 0] == 0
    for y in range(h):
        for x in range(w):
            assert dl[y, x] == py_wta(S[y, x, :min(D - 1, x) + 1], 0.95)
    dr = np.zeros((h, w), np.float32)
    oracle.matchWTARight_SSE(S, dr, w, h, D, 0.95)
    for y in range(h):
        for x in range(w):
            n = min(D - 1, w - 1 - x) + 1
            assert dr[y, x] == py_wta(np.array([S[y, x + d, d] for d in range(n)]), 0.95)


def test_subpixel_and_median():
    S = np.full((1, 16, 8), 40, np.uint16)
    S[0, 5, 2:5] = [30, 10, 20]                     # c0=30,c1=10,c2=20 -> den = c0-c1 = 20 ; 3 + (30-20)/40
    disp = np.zeros((1, 16), np.float32)
    disp[0, 5] = 3
    disp[0, 0] = 3                                  # border column: untouched
    disp[0, 7] = -10                                # invalid: untouched
    oracle.subPixelRefine(S, disp, 16, 1, 8, 0)
    assert disp[0, 5] == np.float32(3 + 10 / 40) and disp[0, 0] == 3 and disp[0, 7] == -10
    rng = np.random.default_rng(5)
    src = rng.uniform(-10, 60, (6, 16)).astype(np.float32)
    dst = np.zeros_like(src)
    oracle.median3x3_SSE(src, dst, 16, 6)
    assert np.array_equal(dst[0], src[0]) and np.array_equal(dst[:, 0], src[:, 0]) and np.array_equal(dst[-1], src[-1])
    for y in range(1, 5):
        for x in range(1, 15):
            assert dst[y, x] == np.sort(src[y - 1:y + 2, x - 1:x + 2].ravel())[4]


def test_gray_pad_speckle():
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], np.uint8)
    assert oracle.rgb2gray(rgb).tolist() == [[76, 150, 29, 255]]
    assert oracle.rgb2gray(rgb, bgr=True).tolist() == [[29, 150, 76, 255]]
    a = np.arange(12, dtype=np.uint8).reshape(3, 4)
    p = oracle.pad_reflect(a, 2, 1, 1, 2)
    assert p.shape == (6, 7) and p[2].tolist() == [0, 0, 1, 2, 3, 3, 2] and p[0].tolist() == p[3].tolist()
    img = np.zeros((20, 30), np.uint8)
    img[2:6, 2:6] = 50                 # 16-pixel speckle -> removed
    img[8:19, 5:29] = 80               # 264-pixel region -> kept
    img[9, 6] = 88                     # within maxDiff of its neighbours: part of the big region
    img[3, 3] = 100                    # differs by 50 from the speckle: its own 1-pixel component
    out = img.copy()
    oracle.filterSpeckles(out, 0, 200, 10)
    assert not out[2:6, 2:6].any() and (out[8:19, 5:29] > 0).all() and out[9, 6] == 88


@pytest.mark.parametrize("D", [16, 64, 192])
def test_avx2_aggregation_twin_is_bit_equal_to_the_scalar_checker(D):
    """`cpu_baseline_simd` times the oracle with its AVX2 aggregation (adds_epu16 / min_epu16: the kind of code the reference's
    SSE natives are): it must be the same function -- every path alone, all eight together, saturating inputs included."""
    rng = np.random.default_rng(D)
    h, w = 9, 23
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for case in range(3):
        if case == 0:
            dsi = rng.integers(0, 25, (h, w, D)).astype(np.uint16)
            p1, p2min, alpha, gamma = 11, 17, 0.5, 35
        elif case == 1:
            dsi = rng.integers(0, 65536, (h, w, D)).astype(np.uint16)          # saturation everywhere
            p1, p2min, alpha, gamma = 40000, 30000, 100.0, 60000
        else:
            dsi = rng.integers(0, 400, (h, w, D)).astype(np.uint16)
            p1, p2min, alpha, gamma = 0, 0, 0.25, 200
        for mask in [1 << k for k in range(8)] + [0xFF, 0x0F, 0xA5]:
            a, b = np.empty_like(dsi), np.empty_like(dsi)
            oracle.aggregate_SSE(img, dsi, a, w, h, D, p1, p2min, alpha, gamma, path_mask=mask)
            oracle.aggregate_SSE(img, dsi, b, w, h, D, p1, p2min, alpha, gamma, path_mask=mask, simd=True)
            assert np.array_equal(a, b), (case, mask)


def test_avx2_wta_twin_is_bit_equal_to_the_scalar_checker():
    """The other half of `cpu_baseline_simd`: left and right winner-take-all with 16 pixels side by side.  Ties (first minimum
    wins, a second minimum equal to the first), all-65535 vectors, widths below D and not multiples of 8 or 16, every uniqueness
    branch (ratio test, neighbour below, neighbour above, invalid)."""
    rng = np.random.default_rng(77)
    seen = set()
    for trial in range(120):
        D = int(rng.choice([8, 16, 24, 48, 64, 192]))
        w, h = int(rng.integers(1, 70)), int(rng.integers(1, 4))
        hi = int(rng.choice([3, 50, 65536]))
        S = rng.integers(0, hi, (h, w, D)).astype(np.uint16)
        if trial % 7 == 0:
            S[:] = 65535
        if trial % 11 == 0:
            S[:] = rng.integers(65530, 65536, (h, w, D)).astype(np.uint16)
        u = float(rng.choice([0.95, 1.0, 0.5, 0.0, 0.77]))
        for fn in (oracle.matchWTA_SSE, oracle.matchWTARight_SSE):
            a, b = np.full((h, w), 7, np.float32), np.full((h, w), 9, np.float32)
            fn(S, a, w, h, D, u)
            fn(S, b, w, h, D, u, simd=True)
            assert np.array_equal(a, b), (trial, D, w, h, hi, u, fn.__name__)
            seen.add(bool((a < 0).any()))
            seen.add(bool((a >= 0).any()))
    assert seen == {True, False}
    S = rng.integers(0, 9, (2, 30, 12)).astype(np.uint16)            # D % 8 != 0: the scalar function runs
    a, b = np.empty((2, 30), np.float32), np.empty((2, 30), np.float32)
    oracle.matchWTA_SSE(S, a, 30, 2, 12, 0.95)
    oracle.matchWTA_SSE(S, b, 30, 2, 12, 0.95, simd=True)
    assert np.array_equal(a, b)


def test_avx2_median_twin_and_the_empty_path_mask():
    rng = np.random.default_rng(12)
    for t in range(60):
        h, w = int(rng.integers(1, 12)), int(rng.integers(1, 40))
        if t % 3 == 0:
            a = rng.integers(-1, 4, (h, w)).astype(np.float32)                     # ties everywhere
        elif t % 3 == 1:
            a = (rng.random((h, w)) * 190).astype(np.float32)
            a[rng.random((h, w)) < 0.3] = -10                                       # the invalid marker among sub-pixel values
        else:
            a = rng.integers(0, 3, (h, w)).astype(np.float32) * 0.5
        x, y = np.full_like(a, 5), np.full_like(a, 6)
        oracle.median3x3_SSE(a, x, w, h)
        oracle.median3x3_SSE(a, y, w, h, simd=True)
        assert np.array_equal(x, y), (t, h, w)
    img = rng.integers(0, 256, (4, 9), dtype=np.uint8)
    dsi = rng.integers(0, 25, (4, 9, 16)).astype(np.uint16)
    a, b = np.full_like(dsi, 3), np.full_like(dsi, 4)
    oracle.aggregate_SSE(img, dsi, a, 9, 4, 16, 11, 17, 0.5, 35, path_mask=0)
    oracle.aggregate_SSE(img, dsi, b, 9, 4, 16, 11, 17, 0.5, 35, path_mask=0, simd=True)
    assert not a.any() and not b.any()


def test_avx2_twin_falls_back_where_it_does_not_apply_and_compute_rsgm_is_unchanged_by_the_flag():
    rng = np.random.default_rng(5)
    h, w, D = 6, 20, 24                                   # D % 16 != 0: the scalar function runs
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    dsi = rng.integers(0, 25, (h, w, D)).astype(np.uint16)
    a, b = np.empty_like(dsi), np.empty_like(dsi)
    oracle.aggregate_SSE(img, dsi, a, w, h, D, 11, 17, 0.5, 35)
    oracle.aggregate_SSE(img, dsi, b, w, h, D, 11, 17, 0.5, 35, simd=True)
    assert np.array_equal(a, b)
    oracle.aggregate_SSE(img, dsi[..., :16].copy(), a[..., :16].copy(), w, h, 16, 11, -5, 0.5, -40, simd=True)   # negative P2: scalar path, no crash
    import synth
    fr = synth.make_frame(40, 96, 64, 0.05, seed=3)
    assert not oracle.get_simd()
    want = oracle.compute_rsgm(fr["left"], fr["left"], fr["right"], dmax=64)
    oracle.set_simd(True)
    try:
        got = oracle.compute_rsgm(fr["left"], fr["left"], fr["right"], dmax=64)
    finally:
        oracle.set_simd(False)
    assert np.array_equal(want, got)
