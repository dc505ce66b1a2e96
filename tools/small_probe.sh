#!/bin/bash
# Usage (GPU box, repo root): tools/small_probe.sh <tag> [B [steps [H W D]]]
# Stand-alone durations of the front/post-stage kernels: rocprofv3 --kernel-trace --stats of tools/small_probe.py (no overlap
# between calls), then every kernel under 400 us with calls, average and share, largest total first.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small_$tag -o $tag -- python3 tools/small_probe.py "$@" > gpurun_out/small_$tag.log 2>&1
python3 - "$(find gpurun_out/small_$tag -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    avg = float(r["AverageNs"]) / 1e3
    if avg < 400:
        tot += float(r["TotalDurationNs"]) / 1e3
        print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} {avg:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  total {float(r['TotalDurationNs'])/1e3:9.1f}")
print("small kernels total us:", round(tot, 1))
PY
