"""Per-step view of a rocprofv3 kernel_stats.csv of a bench.py run: python tools/stats_per_step.py <csv> [replays]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else max(int(r["Calls"]) for r in rows if "gather_fm_fwd" in r["Name"])
tot = 0.0
for r in rows:
    per = float(r["TotalDurationNs"]) / steps / 1e3
    if per < 0.05:
        continue
    tot += per
    print(f"{r['Name'][:70]:70s} calls/step {int(r['Calls']) / steps:5.2f}  avg {float(r['AverageNs']) / 1e3:7.2f} us  per step {per:7.2f} us")
print(f"sum per step {tot:.1f} us over {steps} steps")
