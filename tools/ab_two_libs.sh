#!/bin/bash
# A/B of two builds of the library on ONE box: ab_libs/old.so and ab_libs/new.so (copies of lib/libmi355x_recsys.so built from
# the two trees; ab_libs/ is not tracked) are swapped in turn under the same bench command, three rounds.
#   tools/ab_two_libs.sh [bench.py arguments, default: the C2 step]
L=recsys-benchmark_amd/lib/libmi355x_recsys.so
ARGS=${@:---no-cpu-baseline --no-sweep --no-train-step --no-eager-leg --no-gather-leg}
for i in 1 2 3; do
  for v in old new; do
    cp ab_libs/$v.so $L
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 $ARGS > gpurun_out/ab_$v$i.json 2> gpurun_out/ab_$v$i.err || { tail -3 gpurun_out/ab_$v$i.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/ab_$v$i.json').read().strip().splitlines()[-1]); w=d.get('ms_per_step_windows') or {}; print('$v', d['ms_per_step'], w.get('min'), w.get('max'))"
  done
done
cp ab_libs/new.so $L
