#!/bin/bash
# Usage (GPU box, repo root): tools/other_configs.sh <out.json>   -- bench lines of the other BASELINE shapes
out=$1
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
echo "[" > $out
first=1
run() {
    python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 "$@" > gpurun_out/oc.tmp 2>/dev/null
    [ $first = 1 ] || echo "," >> $out
    first=0
    grep '^{' gpurun_out/oc.tmp | tail -1 >> $out
}
run --shape 375 1242 192 0.05
run --shape 1536 2048 256 0.01 --batch 8
run --shape 540 960 128 0.03
run --shape 540 960 64 0.03
run --batch 1
run --batch 8
run --uniform-random
echo "]" >> $out
python3 - "$out" <<'PY'
import json, sys
for d in json.load(open(sys.argv[1])):
    print(d["config"]["workload"][:70], "|", d["ms_per_step"], "ms |", round(d["value"] / 1e3, 1), "Gdisp/s | b1", d["b1"]["ms_per_frame"])
PY
