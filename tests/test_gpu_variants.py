"""GPU parity of the optional execution variants (environment read at context creation, include/vppx.h "environment"):
the aggregation layouts (VPPX_VERT: 0 eight line-parallel paths, 1 band marching, 3 fused vertical kernel; the default
picks 3 from 3 frames per launch on), sub-stream splitting (VPPX_SUBSTREAMS) and every VPPX_VARIANT token: other
lanes-per-pixel layouts of the line-parallel kernel (gw4 / gw8 / gw16), W / E on the line-parallel kernel (we_line), the
general sum / WTA decision code (sum_general), 8 lanes per pixel in the sum kernel (sum_gl8), the D = 256 ring layouts
(sum_trap0 / sum_trap1) and the one-wave-per-chain maxDistance kernels (maxdist_lds / maxdist_global).  Each variant runs in
a fresh process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import oracle, synth
from vppstereo_amd.engine import Engine
eng = Engine()
for (B, H, W, D) in ((1, 24, 300, 256), (4, 50, 150, 192), (8, 40, 96, 64), (2, 70, 81, 128), (8, 30, 200, 256), (8, 21, 230, 192)):
    b = synth.make_batch(B, H, W, D, 0.05, seed=B * H)
    dev = eng.device
    lv = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev); rv = torch.empty_like(lv)
    out = eng.vpp_rsgm(torch.from_numpy(b["left"]).to(dev), torch.from_numpy(b["right"]).to(dev),
                       torch.from_numpy(b["hints"]).to(dev), l_vpp=lv, r_vpp=rv, seed=11, rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()
    out, lv, rv = out.cpu().numpy(), lv.cpu().numpy(), rv.cpu().numpy()
    for f in range(B):
        oracle.init_rand(11 + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f])
        assert np.array_equal(lo, lv[f]) and np.array_equal(ro, rv[f]), (B, f)
        assert np.array_equal(oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D), out[f]), (B, H, W, D, f)
print("VARIANT_OK", int(eng.uses_vert()))   # how the LAST shape (8 frames, D = 192) was aggregated (8 frames at D = 256: fused only with VPPX_VERT=3)
""" % ROOT


@pytest.mark.parametrize("env", [dict(VPPX_VERT="1"), dict(VPPX_VERT="0"), dict(VPPX_VERT="3"), dict(VPPX_VARIANT="gw16"), dict(VPPX_VARIANT="gw4"),
                                 dict(VPPX_VARIANT="gw8"), dict(VPPX_SUBSTREAMS="2"), dict(VPPX_VERT="1", VPPX_SUBSTREAMS="2"),
                                 dict(VPPX_VERT="3", VPPX_SUBSTREAMS="2"), dict(VPPX_VARIANT="sum_general"), dict(VPPX_VARIANT="sum_gl8"),
                                 dict(VPPX_VARIANT="we_line"), dict(VPPX_VARIANT="we_after"), dict(VPPX_VERT="3", VPPX_VARIANT="sum_trap0"), dict(VPPX_VERT="3", VPPX_VARIANT="sum_trap1,we_line"), dict(VPPX_VARIANT="we_whole"), dict(VPPX_VARIANT="we_lq0"), dict(VPPX_VARIANT="we_lq1"), dict()])
def test_variant_matches_oracle(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "VARIANT_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    # 0 = eight line-parallel paths, 1 = band marching, 3 = fused vertical kernel (the default from 8 frames on)
    want = {"0": 0, "1": 1, "3": 3}.get(env.get("VPPX_VERT"), 3)
    if env.get("VPPX_SUBSTREAMS") == "2" and env.get("VPPX_VERT") != "1":
        want = 0    # sub-stream parts never take the fused kernel (two lock-step launches would race for the same slots)
    if want is not None:
        assert "VARIANT_OK %d" % want in r.stdout, r.stdout[-300:]


@pytest.mark.parametrize("env", [dict(VPPX_VARIANT="maxdist_lds"), dict(VPPX_VARIANT="maxdist_global")])
def test_maxdist_one_wave_kernels_still_match_the_golden_cases(env):
    """The row-wavefront kernel is the default; the LDS-ring and in-place one-wave kernels stay as fall-backs
    (x-descending scans, frames too wide for the LDS ring) and must keep passing the same golden / anchor tests."""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_vpp.py"), "-m", "gpu", "-q", "-x",
                        "-k", "maxdist"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
