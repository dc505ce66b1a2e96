#!/usr/bin/env python3
"""tools/make_pmc_json.py <tag> <out.json> [--batch B] [--step-batch S] [--shape H W D]: per-kernel PMC summary of the passes tools/prof_round.sh left in
gpurun_out/ (run in the build container, where git knows the commit).  Only launches at the profiled batch size are
averaged (the bench also makes B=1 calls): they are told apart by the grid size of the dominant kernels.
FETCH_SIZE is doubled (gfx950 under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md section HBM)."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha)


def passes(tag, name):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{name}", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]   # template arguments kept: variants of a kernel are told apart
            acc[k][r["Counter_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
            try:
                acc[k]["_dur_ns"].append((int(r.get("Grid_Size", 0) or 0), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
            except (KeyError, ValueError):
                pass
    return acc


def big_avg(vals):
    """average over the launches of the timed loop: the grid size that carries the most work (launches x grid).  The bench
    also makes one-frame calls and two unsplit 32-frame calls (stage timing); the loop's 16-frame parts outweigh both."""
    import collections
    by = collections.defaultdict(list)
    for g, v in vals:
        by[g].append(v)
    g = max(by, key=lambda k: k * len(by[k]))
    sel = by[g]
    return sum(sel) / len(sel), len(sel)


def main():
    tag, out = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 16   # frames per launch (a 32-frame step = 2 parts)
    step_batch = int(sys.argv[sys.argv.index("--step-batch") + 1]) if "--step-batch" in sys.argv else 32   # frames per step of the profiled bench run
    parts = int(sys.argv[sys.argv.index("--parts") + 1]) if "--parts" in sys.argv else -(-step_batch // batch)   # launches of a part-sized kernel per step
    HH, WW, DD = ((int(v) for v in sys.argv[sys.argv.index("--shape") + 1:sys.argv.index("--shape") + 4]) if "--shape" in sys.argv else (bench.H, bench.W, bench.D))
    fe, wr, sq = passes(tag, "FETCH_SIZE"), passes(tag, "WRITE_SIZE"), passes(tag, "SQ")
    kern = {}
    # of the template variants of one kernel keep the one that carries the most work (total time in the SQ pass): the timed loop's
    weight = {}
    for k in set(fe) | set(wr) | set(sq):
        src = fe.get(k, {}).get("FETCH_SIZE") or sq.get(k, {}).get("SQ_INSTS_VALU") or wr.get(k, {}).get("WRITE_SIZE") or []
        dur = sq.get(k, {}).get("_dur_ns")
        if dur:   # total time of the variant's launches in the SQ pass
            weight[k] = sum(v for _, v in dur)
            continue
        by = collections.defaultdict(int)
        for g, _ in src:
            by[g] += 1
        weight[k] = max((g * n for g, n in by.items()), default=0)
    best = {}
    for k, w in weight.items():
        short = k.split("<")[0]
        if short not in best or w > weight[best[short]]:
            best[short] = k
    for k in sorted(best.values()):
        d = {}
        if "FETCH_SIZE" in fe.get(k, {}):
            d["FETCH_SIZE_KB"], d["launches"] = big_avg(fe[k]["FETCH_SIZE"])
        if "WRITE_SIZE" in wr.get(k, {}):
            d["WRITE_SIZE_KB"], _ = big_avg(wr[k]["WRITE_SIZE"])
        if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
            d["hbm_GB_per_launch"] = round((2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1e3 / 1e9, 3)
        for c, v in sq.get(k, {}).items():
            if c == "_dur_ns":
                d["kernel_ms_in_SQ_pass"] = big_avg(v)[0] / 1e6
            else:
                d[c], _ = big_avg(v)
        if d.get("GRBM_GUI_ACTIVE") and d.get("kernel_ms_in_SQ_pass"):
            # GRBM_GUI_ACTIVE counts shader-clock cycles per XCD, summed over the 8 XCDs
            d["sclk_mhz_profiled"] = round(d["GRBM_GUI_ACTIVE"] / 8.0 / (d["kernel_ms_in_SQ_pass"] * 1e-3) / 1e6, 1)
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                if c in d:
                    d[c.lower() + "_frac_of_wave_cycles"] = round(d[c] / d["SQ_WAVE_CYCLES"], 3)
        d["variant"] = k
        kern[k.split("<")[0]] = {a: (round(b, 4 if a.startswith('kernel_ms') else (3 if a.endswith('_cycles') or a.startswith('hbm') else 1)) if isinstance(b, float) else b) for a, b in d.items()}
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        commit = None
    # real HBM traffic of one step: every kernel's per-launch bytes x its full-batch launches per step (the sum/WTA kernel
    # runs once per step)
    per_step = None
    ref_n = kern.get("sum_wta_lr_kernel", {}).get("launches")
    if ref_n:
        per_step = round(parts * sum(v["hbm_GB_per_launch"] * v.get("launches", 0) / ref_n for v in kern.values() if "hbm_GB_per_launch" in v), 3)
    doc = {"hbm_GB_per_step": per_step, "note": "rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE and the SQ set, each its own run) of "
                   "`python3 bench.py` by tools/prof_round.sh; per-launch averages over the full-batch launches, summed over the "
                   "chip.  hbm_GB_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB.  SQ_* cycle counters are quad-cycles summed over waves.",
           "commit": commit, "kernel_source_sha": bench.kernel_source_sha(), "batch": batch, "step_batch": step_batch, "parts_per_step": parts, "H": HH, "W": WW, "D": DD,
           "kernels": kern}
    json.dump(doc, open(out, "w"), indent=1)
    for k in ("sgm_vert4_kernel", "sgm_we12_kernel", "sum_wta_lr_kernel"):
        print(k, kern.get(k))


if __name__ == "__main__":
    main()
