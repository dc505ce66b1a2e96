#!/bin/bash
# usage: tools/exp_v3.sh <lib suffix>...   (VPPX_VERT=3 runs of experiment builds under tools/bin/)
for v in "$@"; do
  VPPX_VERT=3 VPPX_VERT_SERIAL=${SERIAL:-1} VPPX_LIB=$PWD/tools/bin/libvppx_$v.so timeout 200 python bench.py --steps 5 --warmup 2 --cpu-frames 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', 'ms/step', d['ms_per_step'], 'agg stage', d['stage_ms']['aggregate_8paths'], 'sum', d['stage_ms']['sum_wta_left'])"
done
