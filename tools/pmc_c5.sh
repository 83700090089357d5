#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-pmc_c5}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pmc in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES"; do
  tag=$(echo $pmc | tr ' ' '+')
  for form in 1 0; do
    MI_SPMM_SLICED=$form timeout -k 10 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p_${form}_$tag -- python3 $R/bench.py --config c5 --steps 3 --warmup 1 --windows 0 --no-cpu-baseline --no-graph > /dev/null 2> $OUT/p_${form}_$tag.err || echo "pass $tag $form failed"
  done
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p_*/*/*_counter_collection.csv"):
    form = f.split("/p_")[1][0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_spmm" in n:
            key = ("sliced" if "k_spmm_sliced" in n else "planned") + "_form" + form
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/spmm_counters.txt", "w") as out:
    for k, cs in sorted(acc.items()):
        line = k + " " + " ".join(f"{c}={sorted(v)[len(v)//2]:.0f}(n={len(v)},max={max(v):.0f})" for c, v in sorted(cs.items()))
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/p_*/
