#!/usr/bin/env python3
"""Does touching the NEXT batch's table rows ~50 us ahead (what a rider in the weight-gradient launch would do) make the
gather + FM forward faster when it runs?  In-graph wall per launch:
    A = [filler GEMM (~45 us, MFMA-bound), gather(batch i)] x copies            rows cold (64 distinct batches, 2.16 GB table)
    B = [prefetch(batch i), filler, gather(batch i)] x copies                    rows touched one filler earlier
    C = [filler] x copies,  D = [prefetch(batch i), filler] x copies             the parts, for the differences
gather cold = A - C, gather after a prefetch = B - D, the prefetch's own cost beside the filler = D - C."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import recsys_benchmark_amd as pkg  # noqa: E402
from recsys_benchmark_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dims, D, B = list(bench.CRITEO_KAGGLE_26), 16, 4096
F = len(dims)
torch.manual_seed(2023)
model = pkg.DeepFM(dims, D, [400, 400, 400], p_dropout=0.5, use_batchnorm=True, embedding_config={"name": "vanilla", "sparse": True},
                   fc_sparse=True).to(dev)
model.pack_tables()
gb = bench.GatherBench(model, B, F, D, dev)
xs = [bench.synth_batch(dims, B, 7000 + i, dev)[0] for i in range(64)]
lib = L.load()
a, b = torch.randn(4096, 1024, device=dev), torch.randn(1024, 4096, device=dev)
c = torch.empty(4096, 4096, device=dev)


def filler():
    torch.mm(a, b, out=c)


def prefetch(i):
    x = xs[i % len(xs)]
    L.check(lib.mi_prefetch_rows(x.data_ptr(), gb.off.data_ptr(), gb.W.data_ptr(), gb.ldw, gb.w1.data_ptr(), gb.ldw1, B, F, gb.N,
                                 L.stream_ptr(dev)), "prefetch")


for _ in range(3):
    filler()
torch.cuda.synchronize()
copies, reps = 32, 20
A = bench.graph_wall_us(lambda i: (filler(), gb.fwd(xs[i % len(xs)], i)), copies, reps, dev)
Bv = bench.graph_wall_us(lambda i: (prefetch(i), filler(), gb.fwd(xs[i % len(xs)], i)), copies, reps, dev)
C = bench.graph_wall_us(lambda i: filler(), copies, reps, dev)
Dv = bench.graph_wall_us(lambda i: (prefetch(i), filler()), copies, reps, dev)
print(f"filler {C:.2f} us | filler + gather {A:.2f} -> gather cold {A - C:.2f} us | prefetch + filler {Dv:.2f} -> prefetch {Dv - C:.2f} us | "
      f"prefetch + filler + gather {Bv:.2f} -> gather after a prefetch {Bv - Dv:.2f} us")
