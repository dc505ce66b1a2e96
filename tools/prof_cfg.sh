#!/bin/bash
# Usage (GPU box, repo root): tools/prof_cfg.sh <tag> [--pmc] [bench args, e.g. --shape 375 1242 192 0.05 --batch 16]
# The profile set of ONE configuration (tools/prof_round.sh is the headline's), each pass its own rocprofv3 run with the
# program directly after `--` and only --kernel-trace next to --pmc (MI355X_MICROARCH.md / gpurun rules):
#   1. --kernel-trace --stats -> gpurun_out/prof_<tag>/ (+ the launch sequence of one period: gpurun_out/period_<tag>.txt)
#   with --pmc also: 2. FETCH_SIZE, 3. WRITE_SIZE, 4. the SQ set (short loops; counter passes serialise the kernels)
tag=$1; shift
pmc=0
if [ "$1" = "--pmc" ]; then pmc=1; shift; fi
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --sustained-steps 0 --no-other-configs --cpu-frames 0 "$@" > gpurun_out/bench_$tag.log 2>&1
echo "trace rc $?"
grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json
python3 tools/trace_summary2.py "$(find gpurun_out/prof_$tag -name '*kernel_trace.csv' | head -1)" > gpurun_out/period_$tag.txt 2>&1
find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1 | xargs -r head -14
# the raw trace is large: keep the stats and the period only
find gpurun_out/prof_$tag -name '*kernel_trace.csv' -size +20M -delete
if [ $pmc = 1 ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o $c -- python3 bench.py --steps 4 --warmup 1 --sustained-steps 0 --no-other-configs --cpu-frames 0 "$@" > gpurun_out/pmc_${tag}_$c.log 2>&1
        echo "pass $c rc $?"
    done
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
        --output-format csv -d gpurun_out/pmc_${tag}_SQ -o SQ -- python3 bench.py --steps 4 --warmup 1 --sustained-steps 0 --no-other-configs --cpu-frames 0 "$@" > gpurun_out/pmc_${tag}_SQ.log 2>&1
    echo "pass SQ rc $?"
fi
