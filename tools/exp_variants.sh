#!/bin/bash
# usage: tools/exp_variants.sh <lib suffix>...   (experiment builds under tools/bin/)
for v in "$@"; do
  VPPX_LIB=$PWD/tools/bin/libvppx_$v.so python bench.py --steps 10 --warmup 3 --cpu-frames 0 > /tmp/exp_$v.json 2>/tmp/exp_$v.err
  python - "$v" <<'PY'
import sys, json
v = sys.argv[1]
try:
    d = json.loads(open(f"/tmp/exp_{v}.json").read().strip().splitlines()[-1])
    print(v, "ms/step", d["ms_per_step"], "agg in-pipeline", d["roofline"]["kernel_ms"], "agg back-to-back", d["roofline"]["kernel_ms_back_to_back"], "sum", d["stage_ms"]["sum_wta_left"])
except Exception as e:
    print(v, "failed", e, open(f"/tmp/exp_{v}.err").read()[-500:])
PY
done
