"""tools/trace_summary.py <kernel_trace.csv> [skip_steps]: per-step timeline of a pipelined loop from a rocprofv3 kernel trace:
for each kernel name the average duration, and per step the launch-stream critical path vs the front stream."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return re.split(r"[<(]", re.sub(r"^void ", "", r["Kernel_Name"]))[0]
# steps are delimited by the sum kernel (one per step)
sums = [i for i, r in enumerate(rows) if nm(r) == "sum_wta_lr_kernel"]
if len(sums) < 6: sys.exit("too few steps")
lo, hi = sums[3], sums[-2]
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
nsteps = sums.index(hi) - sums.index(lo)
print("steps", nsteps, "ms/step", (t1 - t0) / nsteps / 1e6)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    a = acc[nm(r) + " q" + r.get("Queue_Id", "?")]
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:45s} n/step {n / nsteps:5.2f}  ms/step {t / nsteps:7.3f}  avg {t / n:7.3f}")
# one step's timeline
s0 = sums[5]
base = int(rows[s0]["Start_Timestamp"])
print("--- timeline of one step (ms relative to its sum kernel start)")
for r in rows[s0 - 2: sums[6] + 1]:
    print(f"{(int(r['Start_Timestamp']) - base) / 1e6:8.3f} {(int(r['End_Timestamp']) - base) / 1e6:8.3f} q{r.get('Queue_Id','?')} {nm(r)}")
