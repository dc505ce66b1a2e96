#!/bin/bash
# Usage (GPU box, repo root): tools/ab_trees.sh <out.json> <rounds> "<tree_a> <tree_b> ..." [bench args...]
# A/B of whole source trees on ONE box (VERDICT r05 next #1a: the r04 library against the current one).  A tree is a directory
# holding bench.py + a built vppstereo_amd/libvppx.so ("." = this repository; tools/bin/r04tree = `git archive 941b90e` with its
# library built, see docs/NOTEBOOK.md round 6).  Every tree runs its OWN bench.py (the C-ABI grew between rounds), alternated
# <rounds> times, once pipelined and once with --no-pipeline; prints ms_per_step and the sum / WTA stage per run.
out=$1; rounds=$2; trees=$3; shift; shift; shift
cd "${GRAFT_REPO_ROOT:?}"
root=$PWD
mkdir -p gpurun_out
echo "[" > $out
first=1
for r in $(seq 1 $rounds); do
    for t in $trees; do
        for mode in "" "--no-pipeline"; do
            (cd $root/$t && python3 bench.py --no-other-configs --cpu-frames 0 $mode "$@") > gpurun_out/abt.tmp 2> gpurun_out/abt.err
            [ $first = 1 ] || echo "," >> $out
            first=0
            line=$(grep '^{' gpurun_out/abt.tmp | tail -1)
            if [ -z "$line" ]; then line="{\"error\": \"$(tail -3 gpurun_out/abt.err | tr '"\n' "' ")\"}"; fi
            echo "{\"tree\": \"$t\", \"mode\": \"${mode:-pipelined}\", \"round\": $r, \"line\": $line}" >> $out
        done
    done
done
echo "]" >> $out
python3 - "$out" <<'PY'
import json, sys
for d in json.load(open(sys.argv[1])):
    l = d["line"]
    if "error" in l:
        print(d["tree"], d["mode"], d["round"], "ERROR", l["error"][:300]); continue
    st = l.get("stage_ms", {})
    we = l["roofline"].get("other_aggregation_launch_ms") or {}
    print(f'{d["tree"]:22s} {d["mode"]:14s} r{d["round"]} step {l["ms_per_step"]:8.3f}  vert {l["roofline"]["kernel_ms"]:.3f}  '
          f'we {[v.get("in_step") for v in we.values()]}  sum(stage) {st.get("sum_wta_left")}  front {st.get("vpp_rnd")}')
PY
