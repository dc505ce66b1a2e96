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


def _stub_module(dirname, altered):
    """A stand-in `pyrSGM` for the round trip: the oracle's natives under the real module's name (optionally with a cost volume
    that leaves d > x at 0).  The stand-alone script itself imports nothing of this repository; the STUB does, in place of the
    compiled extension a maintainer has."""
    with open(os.path.join(dirname, "pyrSGM.py"), "w") as f:
        f.write("import sys\nsys.path.insert(0, %r)\nimport oracle as _o\n" % ROOT)
        f.write("census5x5_SSE = _o.census5x5_SSE\naggregate_SSE = _o.aggregate_SSE\nmatchWTA_SSE = _o.matchWTA_SSE\n"
                "matchWTARight_SSE = _o.matchWTARight_SSE\nsubPixelRefine = _o.subPixelRefine\nmedian3x3_SSE = _o.median3x3_SSE\n")
        if altered:
            f.write("def costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, w, h, dmax, n=1):\n"
                    "    _o.costMeasureCensus5x5_xyd_SSE(cl, cr, dsi, w, h, dmax, n)\n"
                    "    for x in range(min(w, dmax)):\n        dsi[:, x, x + 1:] = 0\n")
        else:
            f.write("costMeasureCensus5x5_xyd_SSE = _o.costMeasureCensus5x5_xyd_SSE\n")


@pytest.mark.parametrize("altered", [False, True])
def test_dump_and_from_round_trip(tmp_path, altered):
    """--dump DIR -> the stand-alone run_reference.py run where "pyrSGM" exists -> --from DIR: whoever has the submodule needs
    nothing of this repository, whoever has this repository needs nothing of the submodule."""
    d = str(tmp_path / "pin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pin_rsgm.py"), "--dump", d, "--seed", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and os.path.exists(os.path.join(d, "inputs.npz")), r.stdout + r.stderr
    src = open(os.path.join(d, "run_reference.py")).read()
    assert "vppstereo_amd" not in src.split('"""')[2] and "import oracle" not in src   # no repo import in the script's code
    stub = str(tmp_path / "stub")
    os.makedirs(stub)
    _stub_module(stub, altered)
    r = subprocess.run([sys.executable, "run_reference.py", "--pyrsgm-path", stub], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and os.path.exists(os.path.join(d, "outputs.npz")), r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pin_rsgm.py"), "--from", d], capture_output=True, text=True, timeout=300)
    if altered:
        assert r.returncode == 1 and "first differing stage: 4" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 0 and "stage checks agree with the recorded reference outputs" in r.stdout, r.stdout + r.stderr
        assert r.stdout.count("PASS") >= 8
