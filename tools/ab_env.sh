#!/bin/bash
# Usage (GPU box, repo root): tools/ab_env.sh <out.json> "<VAR=a VAR=b ...>" <rounds> [bench args...]
# A/B of environment settings on ONE box: the bench line of every setting, alternated <rounds> times (a box's clocks drift, so
# the settings are interleaved, never run in blocks); prints ms_per_step, the sum / aggregation stage times and the W/E launch.
out=$1; settings=$2; rounds=$3; shift; shift; shift
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
echo "[" > $out
first=1
for r in $(seq 1 $rounds); do
    for s in $settings; do
        env ${s//,/ } python3 bench.py --sustained-steps 0 --no-other-configs --cpu-frames 0 "$@" > gpurun_out/ab.tmp 2> gpurun_out/ab.err
        [ $first = 1 ] || echo "," >> $out
        first=0
        line=$(grep '^{' gpurun_out/ab.tmp | tail -1)
        if [ -z "$line" ]; then line="{\"error\": \"$(tail -3 gpurun_out/ab.err | tr '"\n' "' ")\"}"; fi
        echo "{\"setting\": \"$s\", \"round\": $r, \"line\": $line}" >> $out
    done
done
echo "]" >> $out
python3 - "$out" <<'PY'
import json, sys
for d in json.load(open(sys.argv[1])):
    l = d["line"]
    if "error" in l:
        print(d["setting"], d["round"], "ERROR", l["error"][:300]); continue
    st = l.get("stage_ms", {})
    we = l["roofline"].get("other_aggregation_launch_ms") or {}
    print(f'{d["setting"]:28s} r{d["round"]} step {l["ms_per_step"]:8.3f}  unpiped {l["ms_per_step_unpipelined"]}  vert {l["roofline"]["kernel_ms"]:.3f}  '
          f'we {[v["in_step"] for v in we.values()]}  sum(stage) {st.get("sum_wta_left")}  b1 {l["b1"]["ms_per_frame"]}')
PY
