"""CPU oracle (oracle/vpp_oracle.c) vs the golden vectors produced by the reference's own
Cython build of vpp_core/vpp_core_opt.pyx (tests/golden/make_vpp_golden.py)."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def run_case(impl, case, store):
    """Replay one golden case with an implementation exposing the reference's scan API."""
    inp = case["inp"]
    l = store[f"in_{inp}_l"].copy()
    r = store[f"in_{inp}_r"].copy()
    g = store[f"in_{inp}_g"]
    occ = store[f"in_{inp}_occ"] if case["use_occ"] else np.zeros_like(store[f"in_{inp}_occ"])
    H, W, C = l.shape
    impl.init_rand(case["seed"])
    if case["method"] == "rnd":
        n = impl.virtual_projection_scan_rnd(l, r, g, W, H, C, bool(case["uniform"]), case["wsize"], case["direction"],
                                             case["c"], case["c_occ"], occ, bool(case["discard"]),
                                             bool(case["interpolate"]))
    else:
        n = impl.virtual_projection_scan_max_dist(l, r, g, W, H, C, bool(case["uniform"]), case["wsize"],
                                                  case["agg_x"], case["agg_y"], case["direction"], case["c"],
                                                  case["c_occ"], occ, bool(case["discard"]), bool(case["interpolate"]))
    return n, l, r


@pytest.fixture(scope="module")
def vpp_golden():
    store = np.load(os.path.join(GOLDEN, "vpp_cases.npz"))
    with open(os.path.join(GOLDEN, "vpp_cases.json")) as f:
        cases = json.load(f)
    return store, cases


def test_glibc_rand_matches_fixture():
    with open(os.path.join(GOLDEN, "glibc_rand.json")) as f:
        fix = json.load(f)
    for seed, vals in fix.items():
        got = oracle.rand_stream(int(seed), len(vals))
        assert got.tolist() == vals, f"seed {seed}"


def test_glibc_rand_matches_live_libc():
    libc = ctypes.CDLL("libc.so.6")
    libc.rand.restype = ctypes.c_int
    for seed in (1, 7, 99991):
        libc.srand(ctypes.c_uint(seed))
        want = [libc.rand() for _ in range(2000)]
        assert oracle.rand_stream(seed, 2000).tolist() == want


def test_oracle_matches_reference_cases(vpp_golden):
    store, cases = vpp_golden
    assert len(cases) > 100
    for case in cases:
        n, l, r = run_case(oracle, case, store)
        assert n == case["n_hints"], case
        assert np.array_equal(l, store[case["name"] + "_l"]), case
        assert np.array_equal(r, store[case["name"] + "_r"]), case


def test_golden_cases_cover_every_branch(vpp_golden):
    _, cases = vpp_golden
    for key, vals in dict(method=["rnd", "maxdist"], uniform=[0, 1], direction=[0, 1], interpolate=[0, 1],
                          discard=[0, 1], c_occ=[0.0, 0.3], wsize=[1, 3, 5, 7], use_occ=[0, 1],
                          inp=["rgb", "gray", "border", "dense", "fallback"]).items():
        for v in vals:
            assert any(c[key] == v for c in cases), (key, v)


ANCHOR_SETS = ["vpp_anchors.json", "vpp_anchors_splitmix.json"]


def load_anchors(name):
    """(meta, inputs) of one anchor set.  vpp_anchors.json = SURVEY App. D with numpy's Generator: skipped if this numpy
    draws another stream; vpp_anchors_splitmix.json = the same recipe on the repository's own splitmix64 inputs
    (tests/golden/make_vpp_anchors_splitmix.py): never skips, a hash mismatch of the inputs is a failure."""
    with open(os.path.join(GOLDEN, name)) as f:
        meta = json.load(f)
    if "splitmix" in name:
        inp = synth.anchor_inputs_splitmix(meta["H"], meta["W"], meta["D"], meta["p"])
        assert _sha(inp[0]) == meta["inputs"]["l"] and _sha(inp[1]) == meta["inputs"]["r"]
        assert _sha(inp[2]) == meta["inputs"]["g"] and _sha(inp[4]) == meta["inputs"]["occ1"]
        return meta, inp
    inp = _anchor_inputs(meta)
    if _sha(inp[0]) != meta["inputs"]["l"] or _sha(inp[2]) != meta["inputs"]["g"]:
        pytest.skip("numpy Generator stream differs from the one the App. D anchors were made with "
                    "(the splitmix set covers the same cases)")
    return meta, inp


def _anchor_inputs(meta):
    H, W, D, p = meta["H"], meta["W"], meta["D"], meta["p"]
    rng = np.random.default_rng(0)
    l = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    r = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = np.zeros((H, W), np.float32)
    m = rng.random((H, W)) < p
    g[m] = rng.uniform(1, D - 1, size=m.sum()).astype(np.float32)
    occ0 = np.zeros((H, W), np.uint8)
    occ1 = (rng.random((H, W)) < 0.25).astype(np.uint8)
    return l, r, g, occ0, occ1


@pytest.mark.parametrize("anchors", ANCHOR_SETS)
def test_oracle_matches_full_size_anchors(anchors):
    """540x960 anchors of SURVEY.md App. D (hashes of the reference's outputs)."""
    meta, (l, r, g, occ0, occ1) = load_anchors(anchors)
    H, W = meta["H"], meta["W"]
    for c in meta["cases"]:
        a, b = l.copy(), r.copy()
        if c["name"] == "rnd_occ0":
            oracle.init_rand(1)
            n = oracle.virtual_projection_scan_rnd(a, b, g, W, H, 3, False, 3, 1, 0.4, 0.0, occ0, False, True)
        elif c["name"] == "rnd_occ1":
            oracle.init_rand(1)
            n = oracle.virtual_projection_scan_rnd(a, b, g, W, H, 3, False, 3, 1, 0.4, 0.0, occ1, False, True)
        elif c["name"] == "maxdist_occ1":
            oracle.init_rand(1)
            n = oracle.virtual_projection_scan_max_dist(a, b, g, W, H, 3, False, 3, 64, 3, 1, 0.4, 0.0, occ1, False,
                                                        True)
        elif c["name"] == "rnd_w7_uniform_r2l_cocc":
            oracle.init_rand(3)
            n = oracle.virtual_projection_scan_rnd(a, b, g, W, H, 3, True, 7, 0, 0.4, 0.25, occ1, False, True)
        else:
            continue
        assert n == c["n_hints"]
        assert _sha(a) == c["l"], c["name"]
        assert _sha(b) == c["r"], c["name"]
