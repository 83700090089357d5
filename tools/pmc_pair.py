#!/usr/bin/env python3
"""Eager launches of the three kernels behind bench.py's `roofline` (gather+FM forward; the first input-gradient product
without and with the lookup backward in its epilogue) on cold buffers, for rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_pair.py
tools/pmc_pair_summarise.py turns the passes into the per-launch HBM bytes of the pair (profiles/traffic.json)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import recsys_benchmark_amd as pkg  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dims, D, B = list(bench.CRITEO_KAGGLE_26), 16, 4096
F = len(dims)
torch.manual_seed(2023)
model = pkg.DeepFM(dims, D, [400, 400, 400], p_dropout=0.5, use_batchnorm=True, embedding_config={"name": "vanilla", "sparse": True},
                   fc_sparse=True).to(dev)
model.pack_tables()
gb = bench.GatherBench(model, B, F, D, dev)
xs = [bench.synth_batch(dims, B, 4000 + i, dev)[0] for i in range(48)]
db = bench.DgradBench(B, F, D, 400, dev)
for i in range(48):
    gb.fwd(xs[i], i)
    db.run(i, False)
    db.run(i, True)
torch.cuda.synchronize()
