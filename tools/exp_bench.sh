#!/bin/bash
# Usage (GPU box, repo root, `tools/build_exp.sh` build, VPPX_LIB=tools/bin/libvppx_exp.so): tools/exp_bench.sh <out.jsonl> "<ENV=..>" ["<ENV=..>" ...]
# bench.py (headline loop only) once per environment; one summary line each.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for cfg in "$@"; do
    env $cfg python3 bench.py --steps 20 --warmup 3 --no-other-configs --cpu-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']
print(json.dumps({'env':'$cfg','ms':d['ms_per_step'],'unp':d['ms_per_step_unpipelined'],'vert':k['kernel_ms'],'vert_b2b':k['kernel_ms_back_to_back'],'we':k['other_aggregation_launch_ms'],'b1':d['b1']['ms_per_frame'],'st':d['stage_ms']}))" | tee -a $out
done
