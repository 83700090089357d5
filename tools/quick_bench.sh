#!/bin/bash
# tools/quick_bench.sh <tag> <config> [name-filter] [ENV=..]: one bench.py run without the CPU baseline; prints ms_per_step
# and the eager per-kernel times whose name contains the filter
R=${GRAFT_REPO_ROOT:-.}
tag=$1; cfg=$2; filt=${3:-}; shift; shift; shift
mkdir -p $R/gpurun_out/ab
env "$@" timeout -k 10 240 python $R/bench.py --config $cfg --no-cpu-baseline > $R/gpurun_out/ab/$tag.json 2> $R/gpurun_out/ab/$tag.err || { tail -3 $R/gpurun_out/ab/$tag.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/ab/$tag.json")); k = d.get("kernels") or d.get("kernels_eager_dispatch_clock") or {}
print("$tag", d["ms_per_step"], {n: (k[n]["avg_us"], k[n].get("launches")) for n in k if "$filt" and "$filt" in n})
PY
