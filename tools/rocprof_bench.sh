#!/bin/bash
# rocprofv3 --kernel-trace --stats of ONE bench.py command whose library kernels all run inside replayed hipGraphs
# (no eager per-kernel pass, no gather-only graphs): usage  tools/rocprof_bench.sh <tag> [bench.py flags...]
# -> gpurun_out/<tag>_kernel_stats.csv (+ the bench line in gpurun_out/<tag>.json)
set -o pipefail
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/_prof_$tag -- \
  python3 $R/bench.py --no-cpu-baseline --no-eager-leg --no-gather-leg --no-train-step "$@" > $R/gpurun_out/$tag.json 2> $R/gpurun_out/$tag.err || { tail -5 $R/gpurun_out/$tag.err; exit 1; }
find $R/gpurun_out/_prof_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_kernel_stats.csv \;
TRACE=$(find $R/gpurun_out/_prof_$tag -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open("$TRACE")):
    name = r["Kernel_Name"]
    if "k_empty" in name:
        g[("k_empty", r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(g.items()):
    v.sort()
    print("$tag", k, "n", len(v), "avg %.2f med %.2f min %.2f us" % (sum(v) / len(v), v[len(v) // 2], v[0]))
PY
rm -rf $R/gpurun_out/_prof_$tag
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/${tag}_kernel_stats.csv")))
for r in rows:
    if any(k in r["Name"] for k in ("gather_fm", "k_empty", "slot_fm")):
        print("$tag", r["Name"][:60], "calls", r["Calls"], "avg %.2f min %.2f max %.2f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
