#!/usr/bin/env python3
"""Condense the rocprofv3 runs of tools/collect_profiles.sh into the tracked summaries under profiles/:
   <tag>_<cfg>_kernel_stats.csv   per-kernel Calls / avg / min / max of the kernel trace (graph replays + the eager leg)
   <tag>_pmc_summary.csv          per kernel and counter: mean per launch, and HBM bytes per launch with the gfx950
                                  corrections of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests as 64 B for wide
                                  streaming reads; calibrated per access pattern by tools/pmc_calib.py in round 1)
usage: summarise_profiles.py gpurun_out/<dir> <tag> [out_dir]   (out_dir defaults to profiles/)"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:90]


for cfg in ("c2", "c3", "c5"):
    files = glob.glob(os.path.join(src, f"stats_{cfg}", "*", "*_kernel_trace.csv"))
    if not files:
        continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        by[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = sorted(by.items(), key=lambda kv: -sum(kv[1]))
    with open(os.path.join(out_dir, f"{tag}_{cfg}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "median_us", "min_us", "max_us"])
        for k, v in rows:
            v = sorted(v)
            w.writerow([k, len(v), round(sum(v) / 1e3, 1), round(sum(v) / len(v) / 1e3, 3), round(v[len(v) // 2] / 1e3, 3),
                        round(v[0] / 1e3, 3), round(v[-1] / 1e3, 3)])

pm = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(src, "pmc_*")):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d).split("_")[1]
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            pm[(cfg, short(r["Kernel_Name"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out_dir, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["config", "kernel", "launches", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "TCC_HIT", "TCC_MISS", "L2_hit_rate",
                "TCC_EA0_RDREQ", "TCC_EA0_WRREQ"])
    for (cfg, k), c in sorted(pm.items()):
        def m(name):
            v = c.get(name)
            return sum(v) / len(v) if v else None
        hit, miss = m("TCC_HIT_sum"), m("TCC_MISS_sum")
        n = max(len(v) for v in c.values())
        if n < 3:
            continue
        w.writerow([cfg, k, n, m("FETCH_SIZE"), m("WRITE_SIZE"), hit, miss,
                    round(hit / (hit + miss), 4) if hit is not None and (hit + miss) > 0 else None, m("TCC_EA0_RDREQ_sum"), m("TCC_EA0_WRREQ_sum")])
print("wrote", sorted(os.listdir(out_dir)))
