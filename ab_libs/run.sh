#!/bin/bash
# A/B on ONE box: alternate the two builds of the library
L=recsys-benchmark_amd/lib/libmi355x_recsys.so
for i in 1 2 3; do
  for v in old new; do
    cp ab_libs/$v.so $L
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sweep --no-train-step --no-eager-leg --no-gather-leg > gpurun_out/ab_$v$i.json 2> gpurun_out/ab_$v$i.err || exit 1
    python -c "
import json
d=json.loads(open('gpurun_out/ab_$v$i.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['ms_per_step_windows']['min'], d['ms_per_step_windows']['max'])"
  done
done
cp ab_libs/new.so $L
