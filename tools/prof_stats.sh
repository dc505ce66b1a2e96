#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof_stats.sh <tag> [bench args...]
# Runs bench.py under rocprofv3 --kernel-trace --stats and leaves the CSV summaries in
# gpurun_out/prof_<tag>/ ; copy <tag>_kernel_stats.csv into profiles/ to have it judged.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py "$@" > gpurun_out/bench_$tag.log 2>&1
grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json
find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1 | xargs -r head -40
