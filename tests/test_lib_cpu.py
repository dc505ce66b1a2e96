"""CPU-side checks: the C-ABI library loads and exports every symbol include/vppx.h declares;
without a GPU every compute entry point fails loudly (no CPU fallback); host logic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vppstereo_amd import _lib, dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "vppx.h")).read()
    declared = set(re.findall(r"\b(vppx_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("vppx_ctx")
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/vppx.h but not exported"
    assert set(_lib.EXPORTS) == declared
    assert lib.vppx_version() == 100


def test_param_structs_match_header_defaults():
    p = _lib.vpp_params()
    assert (p.method, p.wsize, p.wsize_agg_x, p.wsize_agg_y, p.direction, p.interpolate) == (0, 3, 64, 3, 1, 1)
    assert abs(p.c - 0.4) < 1e-7 and p.c_occ == 0.0 and p.seed == 1 and p.distance_gamma == 0.3
    r = _lib.rsgm_params()
    assert (r.dmax, r.p1, r.p2min, r.gamma, r.subpixel) == (192, 11, 17, 35, 1)
    assert abs(r.alpha - 0.5) < 1e-7 and abs(r.uniqueness - 0.95) < 1e-7
    assert C.sizeof(_lib.VppxVppParams) == 104 and C.sizeof(_lib.VppxRsgmParams) == 32
    assert (p.bilateral_o_xy, p.bilateral_o_i, p.bilateral_th) == (2.0, 1.0, 0.001)


def test_unknown_parameter_names_are_rejected():
    """ctypes would silently create a Python attribute for a misspelt or reference-style keyword."""
    with pytest.raises(TypeError, match="blending"):
        _lib.vpp_params(blending=0.4)
    with pytest.raises(TypeError, match="wsizeAgg_x"):
        _lib.vpp_params(wsizeAgg_x=64)
    with pytest.raises(TypeError, match="reserved0"):
        _lib.rsgm_params(reserved0=1)
    assert _lib.vpp_params(c=0.25, wsize_agg_x=32).wsize_agg_x == 32


def test_calls_without_a_context_fail_with_a_message():
    lib = _lib.load()
    null = C.c_void_p()
    for rc in (lib.vppx_set_stream_legacy(null), lib.vppx_rand_advance(null, 1), lib.vppx_synchronize(null)):
        assert rc == -7
        assert b"context is NULL" in lib.vppx_last_error()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.vppx_create(C.byref(h), -1)
    assert rc == -7 and not h.value
    assert b"no CPU fallback" in lib.vppx_last_error()
    with pytest.raises(_lib.VppxError):
        _lib.Context()
    import vppstereo_amd
    left = np.zeros((16, 32, 3), np.uint8)
    hints = np.zeros((16, 32), np.float32)
    hints[4, 20] = 3.5
    with pytest.raises(_lib.VppxError):
        vppstereo_amd.vpp(left, left, hints)
    with pytest.raises(Exception):
        vppstereo_amd.compute_rsgm(left, left, left, dmax=64)


def test_vpp_wrapper_early_out_and_shapes():
    # vpp_standalone.py:397-407: copies, gray -> [H,W,1], no hints -> untouched copies (no GPU needed)
    from vppstereo_amd import vpp
    left = np.arange(9 * 14 * 3, dtype=np.uint8).reshape(9, 14, 3)
    lc, rc = vpp(left, left[::-1].copy(), np.zeros((9, 14), np.float64))
    assert lc is not left and np.array_equal(lc, left) and rc.shape == left.shape
    lg, rg = vpp(left[..., 0], left[..., 1], np.zeros((9, 14), np.float32))
    assert lg.shape == (9, 14, 1) and rg.shape == (9, 14, 1)
    with pytest.raises(AssertionError):
        vpp(left, left, np.zeros((9, 14)), method="bogus")


def test_shard_range_partitions_every_frame_once():
    for n in (0, 1, 7, 32, 64, 65):
        for ws in (1, 2, 3, 8):
            seen = []
            for r in range(ws):
                lo, hi = dist.shard_range(n, r, ws)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert dist.frame_seed(0xFFFFFFFF, 2) == 1


def test_gt_reshape_matches_oracle():
    import oracle
    from vppstereo_amd import vpp_core_opt
    rng = np.random.default_rng(3)
    g = np.where(rng.random((7, 9)) < 0.3, rng.uniform(0.5, 9, (7, 9)), 0).astype(np.float32)
    got = vpp_core_opt.gt_reshape(g)
    want = np.empty((63, 4), np.float32)
    n = oracle.lib().vppo_gt_reshape(g.ctypes.data_as(C.POINTER(C.c_float)), 7, 9,
                                     want.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(got, want[:n])


def test_get_seed_is_wall_clock_seconds():
    """vpp_core_opt.pyx:23-28: CLOCK_REALTIME seconds as a float."""
    import time
    from vppstereo_amd import vpp_core_opt
    s = vpp_core_opt.get_seed()
    assert isinstance(s, float) and abs(s - time.time()) < 5.0


def test_every_environment_variable_the_library_reads_is_documented_in_the_header():
    """Hygiene guard: `getenv` appears only in vppx_create (one translation unit), every name it reads is listed in include/vppx.h's
    environment section, and the shipped (non-experiment) build reads at most eight."""
    import re
    csrc = os.path.join(ROOT, "vppstereo_amd", "csrc")
    header = open(os.path.join(ROOT, "include", "vppx.h")).read()
    shipped, experiment = set(), set()
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        depth_exp = 0
        for line in open(os.path.join(csrc, name)):
            s = line.strip()
            if s.startswith("#ifdef VPPX_EXPERIMENT"):
                depth_exp += 1
            elif s.startswith("#endif") and depth_exp:
                depth_exp -= 1
            code = line.split("//")[0]
            for m in re.finditer(r'getenv\("(\w+)"\)', code):
                assert name == "vppx_api.hip", (name, m.group(1))
                (experiment if depth_exp else shipped).add(m.group(1))
    assert shipped and len(shipped) <= 8, shipped
    for v in shipped | experiment:
        assert v in header, v
