#!/bin/bash
# Usage (GPU box, repo root): tools/prof_traffic.sh <out.json> [bench args...]
# HBM traffic per kernel launch from the PMC counters, collected the way MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (kernel-trace only, no other
# trace domain), KB units, FETCH_SIZE doubled for gfx950's wide coalesced reads.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o $c -- python3 bench.py "$@" > gpurun_out/pmc_$c.log 2>&1
done
python3 - "$out" "$*" <<'PY'
import csv, glob, json, re, sys, collections
out, args = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            name = re.split(r"[<(]", name)[0]
            acc[name][c].append(float(r["Counter_Value"]))
kern = {}
for k, d in acc.items():
    fe = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
    wr = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    kern[k] = {"FETCH_SIZE_KB_avg": round(fe, 1), "WRITE_SIZE_KB_avg": round(wr, 1), "launches": len(d["FETCH_SIZE"]),
               "hbm_GB_per_launch_corrected": round((2 * fe + wr) * 1e3 / 1e9, 3)}
doc = {"note": "rocprofv3 --pmc passes (separate runs: FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of `python3 bench.py "
               + args + "`. Units: KB per launch as reported; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide "
               "coalesced reads by 2x on gfx950, so hbm_GB = (2*FETCH_SIZE + WRITE_SIZE)*1e3/1e9. Averages are over all "
               "launches of a kernel in the run (the fused pipeline, the per-kernel timing loop and the stage-timed pass).",
       "batch": int(re.search(r"--batch (\d+)", args).group(1)) if "--batch" in args else 32,
       "kernels": kern}
json.dump(doc, open(out, "w"), indent=1)
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_GB_per_launch_corrected"])[:8]:
    print(k, v)
PY
