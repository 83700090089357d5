#!/bin/bash
# A/B of the weight-gradient multi-GEMM variants on a GPU box: exactness tests, then C2 and C3 steps per variant
# (MI_GEMM_TN_DMA = 0 general 64x64-tile kernel, 1 (default) the LDS-DMA kernel of the weight-gradient form)
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/ab
for v in "$@"; do
  MI_GEMM_TN_DMA=$v timeout -k 10 200 python -m pytest $R/tests/test_gemm_gpu.py -x -q 2>&1 | tail -1
  bash $R/tools/ab_bench.sh v$v MI_GEMM_TN_DMA=$v
  MI_GEMM_TN_DMA=$v timeout -k 10 200 python $R/bench.py --config c3 --no-cpu-baseline > $R/gpurun_out/ab/c3_v$v.json 2>/dev/null
  python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/ab/c3_v$v.json")); print("c3 v$v", d["ms_per_step"], d["roofline"].get("frac"))
PY
done
