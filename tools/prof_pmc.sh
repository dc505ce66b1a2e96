#!/bin/bash
# Usage (GPU box, repo root): tools/prof_pmc.sh <tag> "<counter list>" [bench args...]
# One rocprofv3 --pmc pass (own run, kernel-trace only) ; prints per-kernel averages.
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -o $tag -- python3 bench.py "$@" > gpurun_out/pmc_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, sys, glob, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection.csv")
if not f:
    print("no counter csv", glob.glob(f"gpurun_out/pmc_{tag}/*")); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
