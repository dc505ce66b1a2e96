"""GPU parity, VPP: the HIP path (through the C-ABI, via the drop-in modules) against
 (1) the committed golden vectors of the reference's Cython build, and
 (2) the CPU oracle on fresh seeded inputs, bit-exact (uint8 pattern grid)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
import synth
from test_oracle_vpp import ANCHOR_SETS, load_anchors, run_case

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.fixture(scope="module")
def gpu():
    from vppstereo_amd import vpp_core_opt
    return vpp_core_opt


def test_device_glibc_rand_stream_matches_fixture_and_libc():
    from vppstereo_amd import _lib
    lib, ctx = _lib.load(), _lib.default_context()
    with open(os.path.join(GOLDEN, "glibc_rand.json")) as f:
        fix = json.load(f)
    for seed, vals in fix.items():
        out = np.empty(len(vals), np.int32)
        _lib.check(lib.vppx_rand_stream(ctx.handle, int(seed), 0, len(vals), _lib.np_ptr(out)))
        assert out.tolist() == vals, f"seed {seed}"
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    libc.srand(C.c_uint(77))
    want = [libc.rand() for _ in range(5000)]
    for off, n in ((0, 5000), (1, 100), (991, 2000), (992, 993), (4000, 1000)):
        out = np.empty(n, np.int32)
        _lib.check(lib.vppx_rand_stream(ctx.handle, 77, off, n, _lib.np_ptr(out)))
        assert out.tolist() == want[off:off + n], (off, n)
    # far jump-ahead vs the oracle's sequential generator
    big = oracle.rand_stream(5, 3_000_000)
    out = np.empty(4096, np.int32)
    _lib.check(lib.vppx_rand_stream(ctx.handle, 5, 3_000_000 - 4096, 4096, _lib.np_ptr(out)))
    assert np.array_equal(out, big[-4096:])


def test_rnd_golden_cases_bit_exact(gpu):
    store = np.load(os.path.join(GOLDEN, "vpp_cases.npz"))
    with open(os.path.join(GOLDEN, "vpp_cases.json")) as f:
        cases = [c for c in json.load(f) if c["method"] == "rnd"]
    assert len(cases) > 50
    bad = []
    for case in cases:
        n, l, r = run_case(gpu, case, store)
        ok = n == case["n_hints"] and np.array_equal(l, store[case["name"] + "_l"]) and \
            np.array_equal(r, store[case["name"] + "_r"])
        if not ok:
            bad.append((case["name"], {k: case[k] for k in ("inp", "uniform", "direction", "interpolate", "discard",
                                                           "c_occ", "wsize", "use_occ")},
                        int((l != store[case["name"] + "_l"]).sum()), int((r != store[case["name"] + "_r"]).sum())))
    assert not bad, bad[:8]


def test_maxdist_golden_cases_bit_exact(gpu):
    """virtual_projection_scan_max_dist (vpp_core_opt.pyx:133-341) incl. the n_bins==0 fallback."""
    store = np.load(os.path.join(GOLDEN, "vpp_cases.npz"))
    with open(os.path.join(GOLDEN, "vpp_cases.json")) as f:
        cases = [c for c in json.load(f) if c["method"] == "maxdist"]
    assert len(cases) > 40 and any(c["inp"] == "fallback" for c in cases)
    bad = []
    for case in cases:
        n, l, r = run_case(gpu, case, store)
        ok = n == case["n_hints"] and np.array_equal(l, store[case["name"] + "_l"]) and \
            np.array_equal(r, store[case["name"] + "_r"])
        if not ok:
            bad.append((case["name"], {k: case[k] for k in ("inp", "uniform", "direction", "interpolate", "discard",
                                                           "c_occ", "wsize", "use_occ", "agg_x", "agg_y")},
                        int((l != store[case["name"] + "_l"]).sum()), int((r != store[case["name"] + "_r"]).sum())))
    assert not bad, bad[:8]


def test_maxdist_vs_oracle_random_inputs(gpu):
    rng = np.random.default_rng(77)
    for (H, W, C, p, dmax) in [(30, 90, 3, 0.05, 20.0), (24, 200, 1, 0.03, 120.0), (9, 11, 3, 0.4, 5.0)]:
        for trial in range(4):
            l = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
            r = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
            if trial == 3:  # low-contrast images: many ties and narrow intervals
                l = (l // 64 * 64).astype(np.uint8)
                r = (r // 64 * 64).astype(np.uint8)
            g = np.where(rng.random((H, W)) < p, rng.uniform(0.05, dmax, (H, W)), 0).astype(np.float32)
            g[rng.random((H, W)) < p / 4] = np.float32(rng.integers(1, 9))
            occ = (rng.random((H, W)) < 0.3).astype(np.uint8)
            uniform, direction, interp, discard = [bool(b) for b in rng.integers(0, 2, 4)]
            wsize = int(rng.choice([1, 3, 5]))
            ax, ay = int(rng.choice([64, 9, 130, 1])), int(rng.choice([3, 1, 5]))
            c, c_occ = float(np.float32(rng.uniform(0.05, 0.95))), float(np.float32(rng.choice([0.0, 0.3])))
            a0, b0, a1, b1 = l.copy(), r.copy(), l.copy(), r.copy()
            n0 = oracle.virtual_projection_scan_max_dist(a0, b0, g, W, H, C, uniform, wsize, ax, ay, direction, c, c_occ,
                                                         occ, discard, interp)
            n1 = gpu.virtual_projection_scan_max_dist(a1, b1, g, W, H, C, uniform, wsize, ax, ay, direction, c, c_occ, occ,
                                                      discard, interp)
            cfg = dict(uniform=uniform, direction=direction, interp=interp, discard=discard, wsize=wsize, ax=ax, ay=ay)
            assert n0 == n1, cfg
            assert np.array_equal(a0, a1), (cfg, int((a0 != a1).sum()))
            assert np.array_equal(b0, b1), (cfg, int((b0 != b1).sum()))


@pytest.mark.parametrize("anchors", ANCHOR_SETS)
def test_maxdist_full_size_anchor_hash(gpu, anchors):
    meta, (l, r, g, occ0, occ1) = load_anchors(anchors)
    H, W = meta["H"], meta["W"]
    c = {c["name"]: c for c in meta["cases"]}["maxdist_occ1"]
    a, b = l.copy(), r.copy()
    n = gpu.virtual_projection_scan_max_dist(a, b, g, W, H, 3, False, 3, 64, 3, 1, 0.4, 0.0, occ1, False, True)
    assert n == c["n_hints"] and _sha(a) == c["l"] and _sha(b) == c["r"]


@pytest.mark.parametrize("anchors", ANCHOR_SETS)
def test_rnd_full_size_anchor_hashes(gpu, anchors):
    """540x960x3, 3 % hints: SHA-256 of the reference's outputs (the same recipe on splitmix64 inputs)."""
    meta, (l, r, g, occ0, occ1) = load_anchors(anchors)
    H, W = meta["H"], meta["W"]
    by = {c["name"]: c for c in meta["cases"]}
    for name, occ, args, seed in (("rnd_occ0", occ0, (False, 3, 1, 0.4, 0.0), 1), ("rnd_occ1", occ1, (False, 3, 1, 0.4, 0.0), 1),
                                  ("rnd_w7_uniform_r2l_cocc", occ1, (True, 7, 0, 0.4, 0.25), 3)):
        a, b = l.copy(), r.copy()
        gpu.init_rand(seed)
        n = gpu.virtual_projection_scan_rnd(a, b, g, W, H, 3, args[0], args[1], args[2], args[3], args[4], occ, False, True)
        assert n == by[name]["n_hints"]
        assert _sha(a) == by[name]["l"], name
        assert _sha(b) == by[name]["r"], name


@pytest.mark.parametrize("H,W,C,p,dmax", [(37, 53, 3, 0.07, 20.0), (64, 300, 1, 0.03, 190.0), (5, 7, 3, 0.5, 4.0),
                                          (120, 260, 3, 0.25, 60.0),
                                          # rows of 2048 (the last width whose R lists use 16-bit entries in LDS), wider (32-bit
                                          # entries), and too wide for LDS (lists through memory)
                                          (9, 2048, 3, 0.04, 100.0), (7, 2300, 3, 0.03, 60.0), (6, 3700, 1, 0.05, 40.0)])
def test_rnd_vs_oracle_random_inputs(gpu, H, W, C, p, dmax):
    rng = np.random.default_rng(H * 1000 + W)
    for trial in range(6):
        l = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
        r = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
        g = np.where(rng.random((H, W)) < p, rng.uniform(0.05, dmax, (H, W)), 0).astype(np.float32)
        g[rng.random((H, W)) < p / 4] = np.float32(rng.integers(1, 9))
        occ = (rng.random((H, W)) < 0.3).astype(np.uint8)
        uniform, direction, interp, discard = [bool(b) for b in rng.integers(0, 2, 4)]
        wsize = int(rng.choice([1, 3, 5, 7, 9]))
        c, c_occ = float(np.float32(rng.uniform(0.05, 0.95))), float(np.float32(rng.choice([0.0, 0.2, 0.7])))
        seed = int(rng.integers(0, 2**31))
        a0, b0 = l.copy(), r.copy()
        oracle.init_rand(seed)
        n0 = oracle.virtual_projection_scan_rnd(a0, b0, g, W, H, C, uniform, wsize, direction, c, c_occ, occ, discard, interp)
        a1, b1 = l.copy(), r.copy()
        gpu.init_rand(seed)
        n1 = gpu.virtual_projection_scan_rnd(a1, b1, g, W, H, C, uniform, wsize, direction, c, c_occ, occ, discard, interp)
        cfg = dict(uniform=uniform, direction=direction, interp=interp, discard=discard, wsize=wsize, c=c, c_occ=c_occ)
        assert n0 == n1, cfg
        assert np.array_equal(a0, a1), (cfg, int((a0 != a1).sum()))
        assert np.array_equal(b0, b1), (cfg, int((b0 != b1).sum()))


def test_stream_continues_across_scans_like_libc(gpu):
    """Two scans after one init_rand consume one continuous glibc stream (global libc state
    in the reference)."""
    rng = np.random.default_rng(9)
    H, W = 30, 40
    l = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    r = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = np.where(rng.random((H, W)) < 0.1, rng.uniform(1, 12, (H, W)), 0).astype(np.float32)
    occ = np.zeros((H, W), np.uint8)
    a0, b0, a1, b1 = l.copy(), r.copy(), l.copy(), r.copy()
    oracle.init_rand(11)
    gpu.init_rand(11)
    for _ in range(2):
        oracle.virtual_projection_scan_rnd(a0, b0, g, W, H, 3, False, 3, 1, 0.4, 0.0, occ, False, True)
        gpu.virtual_projection_scan_rnd(a1, b1, g, W, H, 3, False, 3, 1, 0.4, 0.0, occ, False, True)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)


def test_vpp_wrapper_matches_oracle_wrapper():
    """vpp() drop-in (vpp_standalone.py:396) incl. g_occ from the occlusion heuristic and the
    distance-patch option."""
    from vppstereo_amd import vpp_standalone, filter as vfilter
    fr = synth.make_frame(96, 160, 48, 0.05, seed=5)
    _, conf_o = oracle.occlusion_heuristic(fr["hints"])
    _, conf_g = vfilter.occlusion_heuristic(fr["hints"])
    assert np.array_equal(conf_o, conf_g)
    for kw in (dict(), dict(g_occ=conf_o, c_occ=0.1), dict(wsize=5, uniform_color=True, left2right=False),
               dict(wsize=7, use_distance_patch=True, g_occ=conf_o),
               dict(wsize=7, use_bilateral_patch=True),                       # TPAMI config (README.md:434-437)
               dict(wsize=5, use_bilateral_patch=True, bilateral_o_xy=3, bilateral_o_i=12, g_occ=conf_o, c_occ=0.2),
               dict(wsize=5, use_bilateral_patch=True, method="maxDistance", wsizeAgg_x=17)):
        oracle.init_rand(4)
        lo, ro = oracle.vpp(fr["left"], fr["right"], fr["hints"], **kw)
        vpp_standalone.init_rand(4)
        lg, rg = vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"], **kw)
        assert lg.dtype == np.uint8 and lg.shape == lo.shape
        assert np.array_equal(lo, lg) and np.array_equal(ro, rg), kw
    # gray input -> [H,W,1]
    oracle.init_rand(4)
    lo, ro = oracle.vpp(fr["left"][..., 0], fr["right"][..., 0], fr["hints"])
    vpp_standalone.init_rand(4)
    lg, rg = vpp_standalone.vpp(fr["left"][..., 0], fr["right"][..., 0], fr["hints"])
    assert lg.shape == (96, 160, 1) and np.array_equal(lo, lg) and np.array_equal(ro, rg)


def test_occlusion_heuristic_golden():
    from vppstereo_amd import filter as vfilter
    G = np.load(os.path.join(GOLDEN, "glue_cases.npz"))
    for i in range(2):
        _, conf = vfilter.occlusion_heuristic(G[f"occ{i}_in"])
        assert np.array_equal(conf, G[f"occ{i}_conf"]), i


def test_occlusion_heuristic_returns_the_references_pair():
    """Element [0] (filter.py:283-292: filtered hints, un-warped, interpolate_disparity(dmap, 3)) next to the mask, against
    the oracle's restatement, including the unguarded dmap[y, x +- 1] reads at both borders (SURVEY C-8: column -1 wraps
    to W-1, column W is the next row's first pixel)."""
    from vppstereo_amd import filter as vfilter
    rng = np.random.default_rng(5)
    for i, (h, w, p) in enumerate([(24, 64, 0.5), (16, 48, 0.8), (33, 70, 0.3), (135, 240, 0.05)]):
        d = np.zeros((h, w), np.float32)
        m = rng.random((h, w)) < p
        d[m] = (rng.integers(8, 80, int(m.sum())) * 0.125).astype(np.float32)   # small disparities: many land next to each other
        d[h // 3: 2 * h // 3, w // 3: w // 2][d[h // 3: 2 * h // 3, w // 3: w // 2] > 0] += 9.0
        d[:, 0] = np.where(rng.random(h) < 0.5, 0, d[:, 0])       # zeros and values in the border columns
        for kw in (dict(), dict(th_conf=0.5, th_filter=1.5)): # (the second: rejected pixels survive the filter)
            want_d, want_c = oracle.occlusion_heuristic(d, **kw)
            got_d, got_c = vfilter.occlusion_heuristic(d, **kw)
            assert got_d.dtype == np.float32 and got_d.shape == d.shape
            assert np.array_equal(got_c, want_c), (i, kw)
            assert np.array_equal(got_d, want_d), (i, kw, int((got_d != want_d).sum()))
            assert (want_d != 0).any()
    filled = 0
    for _ in range(3):   # dense small maps: the interpolation actually fills something, also across the row ends
        d = (rng.integers(0, 3, (12, 20)) * rng.integers(8, 12, (12, 20)) * 0.125).astype(np.float32)
        want_d, _ = oracle.occlusion_heuristic(d)
        got_d, _ = vfilter.occlusion_heuristic(d)
        assert np.array_equal(got_d, want_d)
        filled += int((want_d != 0).sum())
    assert filled > 0


def test_batched_vpp_is_sharding_independent():
    """Frame f of a batch uses srand(seed+f): the same frames give the same result whatever the
    batch split (what makes frame sharding across GPUs exact)."""
    from vppstereo_amd import _lib
    lib, ctx = _lib.load(), _lib.default_context()
    b = synth.make_batch(5, 40, 64, 24, 0.06, seed=77)

    def run(lo, hi, seed0):
        l = np.ascontiguousarray(b["left"][lo:hi]).copy()
        r = np.ascontiguousarray(b["right"][lo:hi]).copy()
        g = np.ascontiguousarray(b["hints"][lo:hi])
        p = _lib.vpp_params(seed=seed0 + lo)
        nh = (C.c_int64 * (hi - lo))()
        _lib.check(lib.vppx_vpp_host(ctx.handle, C.byref(p), hi - lo, 40, 64, 3, _lib.np_ptr(l), _lib.np_ptr(r),
                                     _lib.np_ptr(g), None, None, nh))
        return l, r, list(nh)
    l_all, r_all, nh = run(0, 5, 1000)
    l_a, r_a, _ = run(0, 2, 1000)
    l_b, r_b, _ = run(2, 5, 1000)
    assert np.array_equal(l_all, np.concatenate([l_a, l_b])) and np.array_equal(r_all, np.concatenate([r_a, r_b]))
    for f in range(5):
        a0, b0 = b["left"][f].copy(), b["right"][f].copy()
        oracle.init_rand(1000 + f)
        n = oracle.virtual_projection_scan_rnd(a0, b0, b["hints"][f], 64, 40, 3, False, 3, 1, 0.4, 0.0,
                                               np.zeros((40, 64), np.uint8), False, True)
        assert n == nh[f] and np.array_equal(a0, l_all[f]) and np.array_equal(b0, r_all[f])


@pytest.mark.parametrize("kw", [dict(wsize=3, interpolate=1), dict(wsize=3, interpolate=0, discard_occluded=0),
                                dict(wsize=5, interpolate=1, direction=0), dict(wsize=7, interpolate=0, uniform_color=1),
                                dict(wsize=1, interpolate=1)])
def test_mixed_density_batch_uses_both_thread_mappings(kw):
    """One batch whose frames run from no hints at all to a hint on every pixel (R lists that overflow,
    windows full of hints), with occluded hints, hints hugging the left border
    (targets < 0, the -1 wraparound of the un-interpolated write) and large disparities."""
    from vppstereo_amd import _lib
    lib, ctx = _lib.load(), _lib.default_context()
    H, W, C3 = 36, 90, 3
    dens = [0.0, 0.004, 0.03, 0.08, 0.1, 0.12, 0.3, 1.0]
    B = len(dens)
    rng = np.random.default_rng(99)
    l = rng.integers(0, 256, (B, H, W, C3), dtype=np.uint8)
    r = rng.integers(0, 256, (B, H, W, C3), dtype=np.uint8)
    g = np.zeros((B, H, W), np.float32)
    occ = np.zeros((B, H, W), np.uint8)
    for f, p in enumerate(dens):
        m = rng.random((H, W)) < p
        g[f][m] = rng.uniform(0.2, 40.0, int(m.sum())).astype(np.float32)
        g[f][:, :3][m[:, :3]] = rng.choice(np.array([0.5, 1.5, 2.49, 2.5, 3.0], np.float32), int(m[:, :3].sum()))
        occ[f] = (rng.random((H, W)) < 0.3) & m
    lg, rg = l.copy(), r.copy()
    p = _lib.vpp_params(seed=321, c_occ=0.15, **kw)
    nh = (C.c_int64 * B)()
    _lib.check(lib.vppx_vpp_host(ctx.handle, C.byref(p), B, H, W, C3, _lib.np_ptr(lg), _lib.np_ptr(rg), _lib.np_ptr(g),
                                 _lib.np_ptr(occ), None, nh))
    for f in range(B):
        a0, b0 = l[f].copy(), r[f].copy()
        oracle.init_rand(321 + f)
        n = oracle.virtual_projection_scan_rnd(a0, b0, g[f], W, H, C3, bool(kw.get("uniform_color", 0)), kw["wsize"],
                                               kw.get("direction", 1), 0.4, 0.15, occ[f],
                                               bool(kw.get("discard_occluded", 0)), bool(kw["interpolate"]))
        assert n == nh[f], (f, kw)
        assert np.array_equal(a0, lg[f]), (f, dens[f], kw)
        assert np.array_equal(b0, rg[f]), (f, dens[f], kw)


@pytest.mark.parametrize("h,w,p", [(61, 97, 0.05), (135, 240, 0.03), (8, 8, 0.5), (40, 700, 0.3), (270, 480, 0.01), (12, 1300, 0.05)])
def test_occlusion_mask_single_launch_kernel(h, w, p):
    """The mask alone (Engine.occlusion_heuristic, what the hot path takes: occ_warp(4)_kernel + occ_test_kernel) against the
    oracle: default and other window / weight / threshold parameters -- windows of 2 to 420 positions, i.e. one to four
    positions per lane and the general loop; a filter threshold that keeps rejected pixels --, dense and sparse hints, rows
    wider than one block, widths that are no multiple of four, several frames per call."""
    import torch
    from vppstereo_amd.engine import Engine
    eng = Engine()
    b = synth.make_batch(3, h, w, 64, p, seed=h + w)
    hints = torch.from_numpy(np.ascontiguousarray(b["hints"])).to(eng.device)
    for kw in (dict(), dict(rx=5, ry=3), dict(rx=13, ry=11, l=1.5, g=0.3), dict(rx=1, ry=1), dict(th_conf=0.25, th_filter=0.1),
               dict(rx=9, ry=7, l=0.5, g=0.9, th_conf=2), dict(rx=17, ry=13, l=0.7), dict(rx=21, ry=19, l=0.4, g=0.5),
               dict(th_conf=0.5, th_filter=1.5)):
        got = eng.occlusion_heuristic(hints, **kw).cpu().numpy()
        for f in range(3):
            want = oracle.occlusion_heuristic(b["hints"][f], **kw)[1]
            assert np.array_equal(got[f], want), (kw, f, int((got[f] != want).sum()))


@pytest.mark.parametrize("maskocc,method", [(True, "rnd"), (False, "rnd"), (True, "maxDistance")])
def test_run_frame_equals_the_three_drop_in_calls(maskocc, method):
    """vppstereo_amd.pipeline.run_frame (one call, one upload: vppx_occ_vpp_rsgm_host) against filter.occlusion_heuristic +
    vpp_standalone.vpp + rsgm.compute_rsgm called one after the other (test.py:154-225), twice in a row: same disparities,
    same patterned pair, same mask, and the same position of the shared random stream afterwards."""
    from vppstereo_amd import filter as vfilter, pipeline, rsgm, vpp_standalone
    D = 64
    fr = synth.make_frame(60, 150, D, 0.05, seed=31)
    kw = dict(wsize=5, blending=0.3, c_occ=0.2, method=method)
    vpp_standalone.init_rand(7)
    want = []
    for _ in range(2):
        conf = vfilter.occlusion_heuristic(fr["hints"])[1] if maskocc else None
        lc, rc = vpp_standalone.vpp(fr["left"], fr["right"], fr["hints"], g_occ=conf, **kw)
        want.append((rsgm.compute_rsgm(fr["left"], lc, rc, dmax=D, p1=9), lc, rc, conf))
    vpp_standalone.init_rand(7)
    for i in range(2):
        d, lc, rc, conf = pipeline.run_frame(fr["left"], fr["right"], fr["hints"], maskocc=maskocc, vpp_kw=kw, rsgm_kw=dict(dmax=D, p1=9),
                                             return_patterns=True)
        assert np.array_equal(d, want[i][0]) and np.array_equal(lc, want[i][1]) and np.array_equal(rc, want[i][2]), i
        if maskocc:
            assert np.array_equal(conf, want[i][3])
    d2 = pipeline.run_frame(fr["left"], fr["right"], np.zeros_like(fr["hints"]), maskocc=maskocc, rsgm_kw=dict(dmax=D))   # no hints at all
    assert np.array_equal(d2, rsgm.compute_rsgm(fr["left"], fr["left"], fr["right"], dmax=D))
    # no hints and use_distance_patch: vpp() returns the untouched pair before dmin / dmax are looked at (vpp_standalone.py:407)
    lc0, rc0 = vpp_standalone.vpp(fr["left"], fr["right"], np.zeros_like(fr["hints"]), use_distance_patch=True)
    assert np.array_equal(lc0, fr["left"]) and np.array_equal(rc0, fr["right"])
    d3 = pipeline.run_frame(fr["left"], fr["right"], np.zeros_like(fr["hints"]), maskocc=maskocc, vpp_kw=dict(use_distance_patch=True),
                            rsgm_kw=dict(dmax=D))
    assert np.array_equal(d3, d2)
