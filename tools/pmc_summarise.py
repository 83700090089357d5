"""Mean / min / max of one rocprofv3 --pmc counter per kernel from the counter_collection CSVs under a directory."""
import csv, glob, sys, collections
d = sys.argv[1]
want = sys.argv[2:] or ["k_gather_fm_fwd", "k_gather_fm_bwd_rows"]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        for w in want:
            if w in name:
                acc[(w, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k},{c},{len(v)},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f}")
