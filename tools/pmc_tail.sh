#!/bin/bash
# SQ counters of the MLP-tail products for one bench variant: tools/pmc_tail.sh <tag> [env assignments...]
set -o pipefail
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  t=$(echo $set | tr ' ' '+')
  env "$@" timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/_$tag.$t -- \
    python3 $R/bench.py --steps 5 --warmup 2 --windows 0 --no-cpu-baseline --no-graph --no-gather-leg --no-train-step --no-sweep --no-eager-leg > /dev/null 2> $R/gpurun_out/pmc/$tag.$t.err || echo "pass $t failed"
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc/_$tag.*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_tail_fwd" in n or "k_tail_dgrad" in n or "k_tail_head" in n or "k_gemm_f32_multi" in n or "k_gemm_tn_multi" in n:
            acc[re.search(r"(k_\\w+(<[^>]*>)?)", n).group(1).replace(",", ";")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$R/gpurun_out/pmc/$tag.summary.csv", "w") as out:
    names = sorted({c for k in acc.values() for c in k})
    out.write("kernel,launches," + ",".join(names) + "\n")
    for k, cs in sorted(acc.items()):
        out.write(k + "," + str(max(len(v) for v in cs.values())) + "," + ",".join(f"{sum(cs[c]) / len(cs[c]):.0f}" if c in cs else "" for c in names) + "\n")
print(open("$R/gpurun_out/pmc/$tag.summary.csv").read())
PY
rm -rf $R/gpurun_out/pmc/_$tag.*
