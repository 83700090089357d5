#!/bin/bash
# quick A/B helper on a GPU box: tools/ab_bench.sh <tag> [env assignments] -> prints ms_per_step and the tail kernels' eager times
R=${GRAFT_REPO_ROOT:-.}
tag=$1; shift
mkdir -p $R/gpurun_out/ab
env "$@" timeout -k 10 240 python $R/bench.py --no-cpu-baseline --no-train-step --no-sweep --no-gather-leg > $R/gpurun_out/ab/$tag.json 2> $R/gpurun_out/ab/$tag.err || { tail -3 $R/gpurun_out/ab/$tag.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/ab/$tag.json")); k = d["kernels_eager_dispatch_clock"]
print("$tag", d["ms_per_step"], d["ms_per_step_windows"]["median"], {n: k[n]["avg_us"] for n in k if "fwd_gemm" in n or "dgrad" in n or "multi" in n})
PY
