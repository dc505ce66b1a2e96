#!/bin/bash
# Usage (GPU box, repo root): tools/trace_kernels.sh <tag> [bench args...]   -> per-kernel durations of the LAST full step
# (rocprofv3 --kernel-trace of bench.py; tools/trace_summary.py prints the launch sequence of one step)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -o $tag -- python3 bench.py "$@" > gpurun_out/trace_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/trace_{tag}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the steps of the timed loop: find launches of the sum kernel with the largest grid
big = max(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) for r in rows if "sum_wta_lr" in r["Kernel_Name"])
idx = [i for i, r in enumerate(rows) if "sum_wta_lr" in r["Kernel_Name"] and int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) == big]
a, b = idx[len(idx) // 2 - 1], idx[len(idx) // 2]      # one period in the middle of the run
t0 = int(rows[a + 1]["Start_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:60]}  grid {int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])}")
PY
