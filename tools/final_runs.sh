#!/bin/bash
# The round's tracked evidence in one GPU call: default bench lines of the three configurations (with cpu_baseline, sweep,
# train_step), the sharded world-1 line, then tools/collect_profiles.sh.  usage: tools/final_runs.sh <tag>
R=${GRAFT_REPO_ROOT:-.}
tag=${1:-r03}
O=$R/gpurun_out/$tag
mkdir -p $O
for cfg in c2 c3 c5; do
  timeout -k 10 400 python $R/bench.py --config $cfg > $O/${tag}_bench_$cfg.json 2> $O/bench_$cfg.err || { tail -5 $O/bench_$cfg.err; exit 1; }
  echo "$cfg done"
done
timeout -k 10 300 python $R/bench.py --sharded --no-cpu-baseline > $O/${tag}_bench_sharded_world1.json 2> $O/bench_sharded.err || tail -5 $O/bench_sharded.err
echo "sharded done"
bash $R/tools/collect_profiles.sh $tag/prof $tag > $O/collect.log 2>&1
tail -2 $O/collect.log
