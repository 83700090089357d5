"""How fast would the LightGCN SpMM be if every gather hit the XCD's L2?  (tools/, not product code)

The Yelp2018-shaped adjacency of bench.py with its column indices folded into the first `fold` rows of X (same rows,
same nnz per row, same edge stream; only where the gathers land changes): fold = 8192 rows = 2 MiB of X fits every
XCD's 4 MiB L2, so the kernel time at that fold is what perfect column-block locality could buy the real graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import yelp_graph  # noqa: E402
from recsys_benchmark_amd import _kernels as K  # noqa: E402

dev = torch.device("cuda:0")
adj = yelp_graph().to(dev)
N, D = adj.shape[0], 64
crow, col, val = adj.crow_indices(), adj.col_indices(), adj.values()
X = torch.randn(N, D, device=dev)
Y = torch.empty(N, D, device=dev)


def time_plan(plan, reps=30):
    def run():
        K._spmm(plan, False, val, X, None, 0, Y, None, None, 0, None, 1.0, D)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(6):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (6 * reps)


print(f"N={N} nnz={col.numel()} D={D}")
for fold in (None, 32768, 16384, 8192, 4096, 1024):
    c = col if fold is None else (col % fold)
    plan = K.CsrPlan(crow, c.contiguous(), (N, N))
    print(f"columns folded into the first {fold if fold else N:>6} rows of X ({(fold or N) * D * 4 / 2**20:6.1f} MiB): "
          f"{time_plan(plan):7.2f} us per SpMM")
