#!/bin/bash
# The round's tracked evidence, two GPU calls (each under gpurun's 20-minute limit):
#   tools/final_runs.sh <tag> bench1  default bench lines of C2 / C3 / C5 (cpu_baseline, sweep, train_step), --infer, the sharded
#                                     world-1 line
#   tools/final_runs.sh <tag> bench2  C4 (1e9 rows), Zipf ids / 39 fields, the -m gpu suite's log
#   tools/final_runs.sh <tag> pair    only the fused pair's PMC passes
#   tools/final_runs.sh <tag> prof    tools/collect_profiles.sh (kernel statistics + PMC passes), the fused pair's PMC passes,
#                                     the two SpMM forms' counters
R=${GRAFT_REPO_ROOT:-.}
tag=${1:-r04}
what=${2:-bench}
O=$R/gpurun_out/$tag
mkdir -p $O
if [ $what = bench1 ]; then
  for cfg in c2 c3 c5; do
    timeout -k 10 400 python $R/bench.py --config $cfg > $O/${tag}_bench_$cfg.json 2> $O/bench_$cfg.err || { tail -5 $O/bench_$cfg.err; exit 1; }
    echo "$cfg done"
  done
  timeout -k 10 300 python $R/bench.py --infer > $O/${tag}_bench_infer.json 2> $O/bench_infer.err || tail -5 $O/bench_infer.err
  timeout -k 10 300 python $R/bench.py --sharded --no-cpu-baseline > $O/${tag}_bench_sharded_world1.json 2> $O/bench_sharded.err || tail -5 $O/bench_sharded.err
elif [ $what = bench2 ]; then
  timeout -k 10 400 python $R/bench.py --c4 --no-cpu-baseline > $O/${tag}_bench_c4_1e9rows_world1.json 2> $O/bench_c4.err || tail -5 $O/bench_c4.err
  timeout -k 10 300 python $R/bench.py --ids zipf --no-cpu-baseline --no-train-step > $O/${tag}_bench_c2_zipf.json 2> $O/bench_zipf.err || tail -5 $O/bench_zipf.err
  timeout -k 10 300 python $R/bench.py --fields 39 --no-cpu-baseline --no-train-step > $O/${tag}_bench_c2_fields39.json 2> $O/bench_f39.err || tail -5 $O/bench_f39.err
  timeout -k 10 300 python $R/bench.py --no-head-loss --no-cpu-baseline --no-train-step --no-sweep > $O/${tag}_bench_c2_labels_withheld.json 2> $O/bench_nohl.err || tail -5 $O/bench_nohl.err
  echo "benches done"
  timeout -k 10 500 python -m pytest $R/tests -q -m gpu > $O/${tag}_pytest_gpu.log 2>&1; tail -3 $O/${tag}_pytest_gpu.log
elif [ $what = pair ]; then
  cd /tmp && export TMPDIR=/tmp
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo $pmc | tr ' ' '+')
    timeout -k 10 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/pair/pass_$t -- python3 $R/tools/pmc_pair.py > /dev/null 2> $O/pair_$t.err || echo "pair pass $t failed"
  done
  python3 $R/tools/pmc_pair_summarise.py $O/pair > $O/${tag}_pair_traffic.txt 2>&1; cat $O/${tag}_pair_traffic.txt
  find $O/pair -name "*_counter_collection.csv" | head -3 | while read f; do head -4 $f | cut -c1-400; done > $O/pair_names.txt
  rm -rf $O/pair
else
  bash $R/tools/collect_profiles.sh $tag/prof $tag > $O/collect.log 2>&1
  tail -2 $O/collect.log
  cd /tmp && export TMPDIR=/tmp
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo $pmc | tr ' ' '+')
    timeout -k 10 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/pair/pass_$t -- python3 $R/tools/pmc_pair.py > /dev/null 2> $O/pair_$t.err || echo "pair pass $t failed"
  done
  python3 $R/tools/pmc_pair_summarise.py $O/pair > $O/${tag}_pair_traffic.txt 2>&1; cat $O/${tag}_pair_traffic.txt
  rm -rf $O/pair
  bash $R/tools/pmc_c5.sh $tag/spmm > $O/pmc_c5.log 2>&1; cp $O/spmm/spmm_counters.txt $O/${tag}_spmm_counters.txt 2>/dev/null
  cd $R && bash $R/tools/pmc_tail.sh $tag > $O/pmc_tail.log 2>&1; cp $R/gpurun_out/pmc/$tag.summary.csv $O/${tag}_tail_sq_counters.csv 2>/dev/null
fi
echo done
