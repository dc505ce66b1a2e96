"""The drop-in boundary from PLAIN C: include/vppx.h compiles as strict C99, a C program links against libvppx.so with nothing but the
header, fails loudly without a GPU (no CPU fallback) -- and, on the GPU box, drives the frame stream (vppx_fstream_*) and gets the
disparities `vppstereo_amd.pipeline.FrameStream` gets for the same frames (tests/c/fstream_demo.c)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "fstream_demo.c")
LIBDIR = os.path.join(ROOT, "vppstereo_amd")


def _build(tmp_path):
    exe = str(tmp_path / "fstream_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lvppx", "-Wl,-rpath," + LIBDIR])
    return exe


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_c99_and_a_c_program_links_and_fails_loudly_without_a_gpu(tmp_path):
    exe = _build(tmp_path)
    if _has_gpu():
        pytest.skip("checks the no-GPU failure mode")
    r = subprocess.run([exe, "2", "32", "64", "64", "2", "7"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and r.stdout == ""


def _lcg_frames(n, H, W):
    """The frames tests/c/fstream_demo.c makes (same LCG, same order of draws)."""
    out = []
    for f in range(n):
        s = (12345 + 977 * f) & 0xFFFFFFFF
        k = H * W * 3
        # x_{i+1} = a x_i + c mod 2^32, vectorised by repeated squaring of the affine map
        def run(s0, m):
            a, c = np.uint64(1664525), np.uint64(1013904223)
            xs = np.empty(m, np.uint64)
            x = np.uint64(s0)
            for i in range(m):
                x = (x * a + c) & np.uint64(0xFFFFFFFF)
                xs[i] = x
            return xs, int(x)
        with np.errstate(over="ignore"):
            a1, s = run(s, k)
            left = (a1 >> np.uint64(24)).astype(np.uint8).reshape(H, W, 3)
            a2, s = run(s, k)
            xs = np.minimum(np.arange(W) + 5, W - 1)
            right = (left[:, xs, :] ^ (a2 >> np.uint64(31)).astype(np.uint8).reshape(H, W, 3)).astype(np.uint8)
            a3, s = run(s, H * W)
        r = a3.reshape(H, W)
        hints = np.where((r >> np.uint64(24)) < 10, 4.0 + ((r >> np.uint64(16)) & np.uint64(3)).astype(np.float32) * 0.5, 0.0).astype(np.float32)
        out.append((np.ascontiguousarray(left), np.ascontiguousarray(right), hints))
    return out


def _fnv1a(b):
    h = 1469598103934665603
    for v in b:
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_program_drives_the_frame_stream_and_gets_the_python_streams_disparities(tmp_path):
    from vppstereo_amd.pipeline import FrameStream
    exe = _build(tmp_path)
    n, H, W, D, batch, seed = 7, 24, 48, 64, 3, 11
    r = subprocess.run([exe, str(n), str(H), str(W), str(D), str(batch), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    got = [line.split() for line in r.stdout.strip().splitlines()]
    assert [int(g[0]) for g in got] == list(range(n))
    frames = _lcg_frames(n, H, W)
    with FrameStream(H, W, 3, batch=batch, seed=seed, maskocc=True, rsgm_kw=dict(dmax=D)) as fs:
        want = list(fs.run(iter(frames)))
    for f in range(n):
        assert int(got[f][1], 16) == _fnv1a(want[f].tobytes()), f
    assert len({g[1] for g in got}) > 1
