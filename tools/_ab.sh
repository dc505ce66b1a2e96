for v in 0 1; do VPPX_SUM_FAST=$v python bench.py --steps 8 --warmup 2 --cpu-frames 0 2>/dev/null > gpurun_out/ab$v.json; python -c "
import json
d=json.load(open('gpurun_out/ab$v.json')); print('fast', $v, d['ms_per_step'], d['stage_ms']['sum_wta_left'], d['roofline']['kernel_ms'])"; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
