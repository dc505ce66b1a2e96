#!/usr/bin/env python3
"""Print a short summary of bench.py's JSON line read from stdin (experiment helper)."""
import json, sys
tag = " ".join(sys.argv[1:])
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(tag, "Mdisp/s", d["value"], "ms/step", d["ms_per_step"], "agg_ms", d["roofline"]["kernel_ms"],
              {k: v for k, v in d["stage_ms"].items() if v > 0.2})
