"""Host logic of the frame stream on the CPU: the staging-copy thread pool (vppstereo_amd/csrc/copy_pool.h, plain C++) built with g++
and run -- once as it ships, once under ThreadSanitizer (data races in the hand-off between the pushing thread and the workers would
be silent corruption of frames)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "copy_pool_harness.cpp")
INC = os.path.join(ROOT, "vppstereo_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("flags,iters", [(["-O2"], "300"), (["-O1", "-g", "-fsanitize=thread"], "120")])
def test_copy_pool_copies_correctly_and_race_free(tmp_path, flags, iters):
    exe = str(tmp_path / "copy_pool_harness")
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-Werror", "-I", INC] + flags + [SRC, "-o", exe])
    r = subprocess.run([exe, iters], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "COPY_POOL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
