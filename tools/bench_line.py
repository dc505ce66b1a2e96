"""tools/bench_line.py < bench.json : the handful of numbers of a bench line that matter while iterating."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
dk = r.get("dominant_kernel", r)   # round 4: the dominant kernel is `roofline` itself
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step", d["ms_per_step"], "unpipelined", d.get("ms_per_step_unpipelined"),
      "frac", (d.get("pipeline_roofline") or r)["frac"], "| kernel_ms", dk.get("kernel_ms"), "b2b", dk.get("kernel_ms_back_to_back"), "we",
      dk.get("other_aggregation_launch_ms"), "| b1", d["b1"]["ms_per_frame"], "| no-occ", d.get("ms_per_step_without_g_occ"))
print("   stages", d["stage_ms"])
for o in d.get("other_configs") or []:
    print("   ", o)
