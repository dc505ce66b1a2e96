#!/bin/bash
# Usage (GPU box, repo root): tools/trace_kernels.sh <tag> [bench args...]
# rocprofv3 --kernel-trace of bench.py, then the launch sequence of one period of the timed loop (start offset, duration,
# kernel, grid, queue): tools/trace_summary2.py <csv> [period index]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -o $tag -- python3 bench.py "$@" > gpurun_out/trace_$tag.log 2>&1
python3 tools/trace_summary2.py "$(find gpurun_out/trace_$tag -name '*kernel_trace.csv' | head -1)"
