#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_cmd.sh <tag> "<counter list>" <script.py> [args...]
# One rocprofv3 --pmc pass (own run, kernel-trace only, the program directly after `--`) of any python script; prints
# per-kernel averages of every counter (kernels with at least 3 launches, longest names cut).
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -o $tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, sys, glob, collections, json
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv", glob.glob(f"gpurun_out/pmc_{tag}/*")); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    n = len(next(iter(d.values())))
    if n < 3: continue
    out[k] = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
    out[k]["n"] = n
json.dump(out, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
for k, d in out.items():
    if "sgm_" in k or "sum_wta" in k: print(k, d)
PY
