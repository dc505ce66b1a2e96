#!/bin/bash
# Usage (build container, repo root, after `tools/prof_round.sh r06` and the three `tools/prof_cfg.sh` calls of README.md ran on a GPU box):
# tools/collect_profiles.sh <round tag, e.g. r06>  -- copies the summaries from gpurun_out/ into profiles/ and writes the PMC jsons.
r=$1
set -e
cp "$(find gpurun_out/prof_$r -name '*kernel_stats.csv' | head -1)" profiles/${r}_kernel_stats.csv
cp gpurun_out/bench_$r.json profiles/${r}_hl_bench_under_rocprof.json
[ -f gpurun_out/period_$r.txt ] && cp gpurun_out/period_$r.txt profiles/${r}_hl_period.txt
python tools/make_pmc_json.py $r profiles/${r}_pmc.json --batch 16 --step-batch 32
for cfg in "c3b16 cfg3_b16 375 1242 192 16 16" "c3b32 cfg3_b32 375 1242 192 16 32" "c5b8 cfg5_b8 1536 2048 256 8 8"; do
    set -- $cfg
    cp "$(find gpurun_out/prof_$1 -name '*kernel_stats.csv' | head -1)" profiles/${r}_$2_kernel_stats.csv
    cp gpurun_out/bench_$1.json profiles/${r}_$2_bench_under_rocprof.json
    cp gpurun_out/period_$1.txt profiles/${r}_$2_period.txt
    python tools/make_pmc_json.py $1 profiles/${r}_$2_pmc.json --batch $6 --step-batch $7 --shape $3 $4 $5
done
ls -la profiles | grep " ${r}_"
