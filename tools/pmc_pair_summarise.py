#!/usr/bin/env python3
"""usage: pmc_pair_summarise.py <dir with pass_* rocprofv3 outputs of tools/pmc_pair.py>  -> prints per-kernel means and the
pair's HBM bytes per launch: forward (FETCH as tallied for 64-B sector requests of random rows + the id stream doubled, see
profiles/traffic.json's note) + [epilogue product - plain product] (streams: FETCH doubled) + WRITE exact."""
import collections
import csv
import glob
import os
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "pass_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"k_tail_dgrad<([^>]*)>", n)          # template arguments <DZ, MID, MERGE, FM>: the last one marks the epilogue
        key = ("gather_fm_fwd" if "k_gather_fm_fwd" in n else None if not m else
               "dgrad_fm" if m.group(1).replace(" ", "").split(",")[-1] in ("true", "1") else "dgrad_plain")
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: sum(v[8:]) / max(1, len(v[8:])) for c, v in cs.items()}      # (skip the first launches: cold code / TLB)
    print(k, {c: round(x) for c, x in out[k].items()}, "launches", {c: len(v) for c, v in cs.items()})
if all(k in out for k in ("gather_fm_fwd", "dgrad_fm", "dgrad_plain")):
    f, a, b = out["gather_fm_fwd"], out["dgrad_fm"], out["dgrad_plain"]
    fwd = f.get("FETCH_SIZE", 0) * 1024 + 425984 + f.get("WRITE_SIZE", 0) * 1024
    epi = 2 * (a.get("FETCH_SIZE", 0) - b.get("FETCH_SIZE", 0)) * 1024 + (a.get("WRITE_SIZE", 0) - b.get("WRITE_SIZE", 0)) * 1024
    print("pair bytes per launch: forward", round(fwd), "+ epilogue delta", round(epi), "=", round(fwd + epi))
