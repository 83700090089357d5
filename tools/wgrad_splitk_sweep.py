#!/usr/bin/env python3
"""The tail's three weight gradients (dW_l = dz_l^T a_{l-1}: 400 x 416, 400 x 400, 400 x 400 over K = 4096) as ONE multi-problem
launch (k_gemm_tn_multi), in-graph wall per launch against the K-slice count given to every problem (0 = the library's cut)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from recsys_benchmark_amd import _kernels as K_  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, widths = 4096, [416, 400, 400, 400]
g = torch.Generator().manual_seed(17)
dz = [torch.randn(B, n, generator=g).to(dev) for n in widths[1:]]
act = [torch.randn(B, k, generator=g).to(dev) for k in widths[:-1]]
dW = [torch.zeros(n, k, device=dev) for k, n in zip(widths[:-1], widths[1:])]
for sk in (0, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16):
    probs = [dict(A=dz[l], B=act[l], C=dW[l], M=widths[l + 1], N=widths[l], K=B, lda=widths[l + 1], ldb=widths[l], ldc=widths[l],
                  **({"splitk": sk} if sk else {})) for l in range(3)]
    kind, wgs, cut = K_.gemm_multi_plan(probs, transA=True)
    us = bench.graph_wall_us(lambda i: K_.gemm_multi(probs, transA=True), 16, 10, dev)
    print(f"splitk {sk:2d}: kernel kind {kind}, {wgs:5d} workgroups, slices {cut}: {us:7.2f} us = {3.98e9 / us / 1e6:5.1f} TFLOP/s", flush=True)
