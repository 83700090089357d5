#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers on a GPU box (run from the repo root through gpurun):
#   kernel-trace statistics of the three bench configurations, and FETCH_SIZE / WRITE_SIZE / TCC hit-miss counter passes
#   (separate --pmc passes, as MI355X_MICROARCH.md prescribes) for the dominant kernels.  Outputs under gpurun_out/$1/.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel statistics: IN-GRAPH launches only (no eager per-kernel pass, no gather-only graphs, no extra legs), so that a
# kernel's average is its average inside the replayed step
for cfg in c2 c3 c5; do
  extra=""; [ $cfg = c2 ] && extra="--no-eager-leg --no-gather-leg --no-train-step --no-sweep"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -- python3 $R/bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline $extra > $OUT/stats_$cfg.json 2> $OUT/stats_$cfg.err || exit 1
done
for cfg in c2 c3 c5; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    tag=$(echo $pmc | tr ' ' '+')
    timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_${cfg}_$tag -- python3 $R/bench.py --config $cfg --steps 5 --warmup 2 --windows 0 --no-cpu-baseline --no-graph --no-gather-leg --no-train-step --no-sweep --no-eager-leg > $OUT/pmc_${cfg}_$tag.json 2> $OUT/pmc_${cfg}_$tag.err || echo "pmc $cfg $tag failed"
  done
done
python3 $R/tools/summarise_profiles.py $OUT ${2:-r04} $OUT/summary
# raw traces are tens of MB: keep only the summaries (gpurun copies back at most 64 MiB)
rm -rf $OUT/stats_c?/ $OUT/pmc_*/
echo done
