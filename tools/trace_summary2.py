"""tools/trace_summary2.py <kernel_trace.csv>: the launch sequence of one period (between two sum/WTA launches of the largest
grid, in the middle of the run) of a rocprofv3 --kernel-trace of bench.py: start offset, duration, kernel."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
gs = lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
big = max(gs(r) for r in rows if "sum_wta_lr" in r["Kernel_Name"])
idx = [i for i, r in enumerate(rows) if "sum_wta_lr" in r["Kernel_Name"] and gs(r) == big]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 3
a, b = idx[k], idx[k + 1]
t0 = int(rows[a + 1]["Start_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:56]:56s} grid {gs(r)} q{r['Queue_Id']}")
