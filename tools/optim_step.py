"""Time the row-sparse optimizer step on a Criteo-shaped row-form gradient (B*F rows, skewed duplicates).

    python tools/optim_step.py [--batch 4096] [--iters 50]

Prints ms per step of optim.SparseAdam (sort + one HIP kernel), of the sort alone, of torch.optim.SparseAdam
on the same gradient, and of optim.SparseSGD.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from recsys_benchmark_amd import optim as rbo  # noqa: E402
from bench import CRITEO_KAGGLE_26, synth_batch  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dim", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dims, D = list(CRITEO_KAGGLE_26), a.dim
    N = sum(dims)
    x, _ = synth_batch(dims, a.batch, 1, dev)
    offs = torch.tensor([0] + dims[:-1], device=dev).cumsum(0)
    rows2d = x + offs
    rows = rows2d.reshape(-1)
    vals = torch.randn(rows.numel(), D, device=dev) * 1e-3

    def grad():
        return torch.sparse_coo_tensor(rows.unsqueeze(0), vals, (N, D))

    out = {}
    for name, make in (("optim.SparseAdam", lambda p: rbo.SparseAdam([p], lr=1e-3)),
                       ("torch.optim.SparseAdam", lambda p: torch.optim.SparseAdam([p], lr=1e-3)),
                       ("optim.SparseSGD", lambda p: rbo.SparseSGD([p], lr=1e-2))):
        p = torch.nn.Parameter(torch.zeros(N, D, device=dev))
        opt = make(p)
        p.grad = grad()
        out[name] = timeit(opt.step, a.iters)
        del opt, p
        torch.cuda.empty_cache()
    out["torch.sort(rows) alone"] = timeit(lambda: torch.sort(rows), a.iters)
    from recsys_benchmark_amd import _kernels
    _kernels.note_field_layout(rows2d, offs, N)           # what the multi-field lookup does for its ids
    out["field sort (LDS) alone"] = timeit(lambda: rbo.sort_rows(rows, N), a.iters)
    p = torch.nn.Parameter(torch.zeros(N, D, device=dev))
    opt = rbo.SparseAdam([p], lr=1e-3)
    p.grad = grad()
    out["optim.SparseAdam, field sort"] = timeit(opt.step, a.iters)
    uniq = torch.unique(rows).numel()
    print(f"n={rows.numel()} rows, {uniq} distinct, D={D}, table {N} rows")
    for k, v in out.items():
        print(f"  {k:28s} {v*1e3:9.1f} us/step")


if __name__ == "__main__":
    main()
