#!/bin/bash
# Usage (GPU box, repo root): tools/prof_round.sh <tag>
# The profile set of a round, each pass its own rocprofv3 run with the program directly after `--` and only
# --kernel-trace next to --pmc (MI355X_MICROARCH.md / gpurun rules):
#   1. --kernel-trace --stats of the headline bench loop  -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv (+ the trace)
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE              -> HBM bytes (separate passes: TCC slots)
#   4. --pmc SQ_* (instruction issue / wait split)        -> where the waves' time goes
# The PMC passes run a short loop (counter passes serialise the kernels and replay them); every pass is bounded.
# tools/make_pmc_json.py (run where git is available) turns 2-4 into profiles/<round>_pmc.json.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --sustained-steps 0 --no-other-configs --cpu-frames 0 > gpurun_out/bench_$tag.log 2>&1
grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o $c -- python3 bench.py --steps 4 --warmup 1 --sustained-steps 0 --no-other-configs --cpu-frames 0 > gpurun_out/pmc_${tag}_$c.log 2>&1
    echo "pass $c rc $?"
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/pmc_${tag}_SQ -o SQ -- python3 bench.py --steps 4 --warmup 1 --sustained-steps 0 --no-other-configs --cpu-frames 0 > gpurun_out/pmc_${tag}_SQ.log 2>&1
echo "pass SQ rc $?"
find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1 | xargs -r head -12
