#!/bin/bash
# C3 step against the workgroup cap of k_cross_bwd_head (MI_CROSS_BWD_GRID): every workgroup ends with one atomic per column on
# the same N words
for r in 1 2; do for g in 512 192 128 96 64; do
MI_CROSS_BWD_GRID=$g timeout -k 10 200 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/cg_$g.json 2> gpurun_out/cg_$g.err || { tail -3 gpurun_out/cg_$g.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/cg_$g.json').read().strip().splitlines()[-1]); k=d.get('kernels') or {}; print('grid cap $g', d['ms_per_step'], (k.get('cross_bwd_head') or {}).get('avg_us'))"
done; done
