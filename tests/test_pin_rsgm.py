"""CPU: tests/pin_rsgm.py (the one-command pin of the rSGM natives for whoever has the reference's un-vendored pyrSGM /
OpenCV, /root/reference/.gitmodules:1-3, models/rsgm/rsgm.py:6) -- it must skip cleanly where neither exists, run all stage
checks against a module with the natives' names, and name the FIRST stage that differs."""
import os
import subprocess
import sys
import types

import numpy as np

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pin_rsgm  # noqa: E402


def test_skips_cleanly_without_the_natives():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pin_rsgm.py"), "--pyrsgm-module", "no_such_pyrSGM_module"],
                       capture_output=True, text=True, timeout=120)
    try:
        import cv2  # noqa: F401
        have_cv2 = True
    except Exception:  # noqa: BLE001
        have_cv2 = False
    assert r.returncode == 0, r.stdout + r.stderr
    if not have_cv2:
        assert "skipped: nothing to compare with" in r.stdout


def test_all_stages_run_and_pass_against_a_module_with_the_natives():
    lines = []
    ran, failed = pin_rsgm.run(oracle, None, {"oracle (CPU port)": oracle}, seed=3, log=lines.append)
    assert failed is None and ran == 8          # stages 3-10 (1, 2, 11 need cv2)
    assert sum("PASS" in l for l in lines) == 8 and not any("FAIL" in l for l in lines)


def test_first_differing_stage_is_named():
    """A stand-in "real" module that leaves the d > x cells of the cost volume at 0 and also
    has another median: stage 4 must be reported as the first, with the place to change."""
    fake = types.SimpleNamespace(**{n: getattr(oracle, n) for n in ("census5x5_SSE", "aggregate_SSE", "matchWTA_SSE", "matchWTARight_SSE",
                                                                    "subPixelRefine")})

    def cost(cl, cr, dsi, w, h, dmax, nthreads=1):
        oracle.costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, w, h, dmax, nthreads)
        for x in range(min(w, dmax)):
            dsi[:, x, x + 1:] = 0

    def median(src, dst, w, h):
        oracle.median3x3_SSE(src, dst, w, h)
        dst[:, 0] = 0
    fake.costMeasureCensus5x5_xyd_SSE, fake.median3x3_SSE = cost, median
    lines = []
    ran, failed = pin_rsgm.run(fake, None, {"oracle (CPU port)": oracle}, seed=1, log=lines.append)
    assert failed == 4
    txt = "\n".join(lines)
    assert "first differing stage" in txt and "INVALID_DISP_COST" in txt
    assert any("stage 10" in l and "FAIL" in l for l in lines)     # later differences are still listed
    assert any("stage  3" in l and "PASS" in l for l in lines)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_hip_drop_in_passes_every_stage_check():
    """--gpu side: the HIP natives (vppstereo_amd.pyrSGM, the module rsgm.py:6 would import) through the same stage checks,
    the oracle standing in for the real extension."""
    from vppstereo_amd import pyrSGM as hip_natives
    lines = []
    ran, failed = pin_rsgm.run(oracle, None, {"vppstereo_amd.pyrSGM": hip_natives}, seed=5, log=lines.append)
    assert failed is None and ran == 8, "\n".join(lines)
