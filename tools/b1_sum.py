"""tools/b1_sum.py <kernel_trace.csv>: the launches between the 21st and the 22nd 8-path aggregation launch of a trace of
tools/b1_probe.py: start offset, duration, kernel, queue."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgm_paths_kernel" in r["Kernel_Name"]]
a, b = idx[20], idx[21]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:50]:50s} q{r['Queue_Id']}")
