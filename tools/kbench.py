#!/usr/bin/env python3
"""Stand-alone timing of the gather+FM kernels (dispatch begin/end events) at several batch
sizes on the C2 table.  GPU box only.  python tools/kbench.py [--reps 50] [--zipf]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recsys_benchmark_amd as pkg  # noqa: E402,F401
from recsys_benchmark_amd import _lib  # noqa: E402
from recsys_benchmark_amd.profiling import KernelTimer  # noqa: E402
from bench import CRITEO_KAGGLE_26, alg_bytes_per_sample  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--zipf", action="store_true")
    ap.add_argument("--batches", type=int, nargs="*", default=[4096, 16384, 65536, 262144])
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--dense", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda")
    dims = CRITEO_KAGGLE_26 if a.fields == 26 else [50] * 13 + CRITEO_KAGGLE_26
    F, D, N = len(dims), 16, sum(dims)
    lib = _lib.load()
    gen = torch.Generator().manual_seed(1)
    W = torch.rand(N, D, device=dev) - 0.5
    w1 = torch.randn(N, device=dev)
    bias = torch.zeros(1, device=dev)
    off = torch.tensor([0] + dims[:-1]).cumsum(0).to(dev)
    err = _lib.err_word(dev)
    fb, bb = alg_bytes_per_sample(F, D)
    if a.dense:
        gW = torch.zeros(N, D, device=dev)
        gw1 = torch.zeros(N, device=dev)
    for B in a.batches:
        cols = []
        for d in dims:
            if a.zipf:
                cols.append((d * torch.rand(B, generator=gen).pow(4)).long().clamp_(max=d - 1))
            else:
                cols.append(torch.randint(0, d, (B,), generator=gen))
        x = torch.stack(cols, 1).to(dev)
        # rotate 8 distinct id batches: re-using one batch keeps its rows Infinity-Cache resident and
        # flatters the forward by ~35 % (tools/gather_probe.py)
        xs = [x] + [torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1).to(dev) for _ in range(7)]
        it = [0]
        emb = torch.empty(B, F, D, device=dev)
        yfm = torch.empty(B, device=dev)
        rows = torch.empty(B, F, dtype=torch.int64, device=dev)
        g_emb = torch.randn(B, F, D, device=dev)
        g_y = torch.randn(B, device=dev)
        gvals = torch.empty(B * F, D, device=dev)
        g1 = torch.empty(B * F, device=dev)
        s = _lib.stream_ptr(dev)

        def fwd():
            x = xs[it[0] % len(xs)] if not a.zipf else xs[0]
            it[0] += 1
            _lib.check(lib.mi_gather_fm_fwd(x.data_ptr(), off.data_ptr(), W.data_ptr(), w1.data_ptr(), bias.data_ptr(),
                                            emb.data_ptr(), yfm.data_ptr(), rows.data_ptr(), B, F, D, N, err.data_ptr(), s))

        def bwd():
            if a.dense:
                _lib.check(lib.mi_gather_fm_bwd_dense(rows.data_ptr(), emb.data_ptr(), g_y.data_ptr(), g_emb.data_ptr(),
                                                      gW.data_ptr(), gw1.data_ptr(), None, B, F, D, N, s))
            else:
                _lib.check(lib.mi_gather_fm_bwd_rows(emb.data_ptr(), g_y.data_ptr(), g_emb.data_ptr(), gvals.data_ptr(),
                                                     g1.data_ptr(), None, B, F, D, s))

        for _ in range(5):
            fwd(); bwd()
        torch.cuda.synchronize()
        with KernelTimer(4 * a.reps + 8) as kt:
            for _ in range(a.reps):
                fwd(); bwd()
            torch.cuda.synchronize()
        for k, st in kt.summary().items():
            nb = (fb if "fwd" in k else bb) * B
            print(f"B={B:7d} {k:22s} avg {st['avg_us']:8.2f} us  min {st['min_us']:8.2f} us  "
                  f"{nb/st['avg_us']/1e3:8.1f} GB/s avg  {nb/st['min_us']/1e3:8.1f} GB/s best  ({nb/1e6:.1f} MB)")


if __name__ == "__main__":
    main()
