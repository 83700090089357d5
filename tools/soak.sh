#!/bin/bash
# tools/soak.sh <runs>: the whole GPU suite <runs> times back to back (stops at the first failing run); logs under gpurun_out/soak/
R=${GRAFT_REPO_ROOT:-.}
n=${1:-5}
mkdir -p $R/gpurun_out/soak
for i in $(seq 1 $n); do
  timeout -k 10 600 python -m pytest $R/tests -m gpu -x -q -p no:cacheprovider > $R/gpurun_out/soak/run_$i.log 2>&1 || { echo "run $i FAILED"; tail -30 $R/gpurun_out/soak/run_$i.log; exit 1; }
  echo "run $i: $(tail -1 $R/gpurun_out/soak/run_$i.log)"
done
