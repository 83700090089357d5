#!/usr/bin/env python3
"""Secondary-workload kernel timings (dispatch begin/end events): LightGCN CSR SpMM at the
Yelp2018 shape (C5), the fp32 MFMA GEMM at the CrossNet shapes (C3), DCN-Mix head fwd+bwd, and
the compressed-embedding gathers.  GPU box only:  python tools/kbench2.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recsys_benchmark_amd as pkg  # noqa: E402
from recsys_benchmark_amd import _kernels  # noqa: E402
from recsys_benchmark_amd.layer_dcn import DCN_MixHead, DCNHead  # noqa: E402
from recsys_benchmark_amd.profiling import KernelTimer  # noqa: E402

dev = torch.device("cuda")


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with KernelTimer(64 * reps + 64) as kt:
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
    return kt.summary(), wall


def yelp_graph(U=31668, I=38048, nnz=1128375, seed=2023):
    gen = torch.Generator().manual_seed(seed)
    u = torch.randint(0, U, (nnz,), generator=gen)
    i = (I * torch.rand(nnz, generator=gen).pow(2)).long().clamp_(max=I - 1)
    n = U + I
    idx = torch.stack([torch.cat([u, i + U]), torch.cat([i + U, u])])
    adj = torch.sparse_coo_tensor(idx, torch.ones(2 * nnz), size=(n, n)).coalesce()
    deg = torch.sparse.sum(adj, dim=1).to_dense().clamp_(min=1).pow(-0.5)
    ii = adj.indices()
    return torch.sparse_coo_tensor(ii, adj.values() * deg[ii[0]] * deg[ii[1]], size=(n, n)).coalesce().to_sparse_csr()


def main():
    # ---- C5: LightGCN propagation
    adj = yelp_graph().to(dev)
    N, D, L = adj.shape[0], 64, 3
    nnz = adj.values().numel()
    plan = _kernels.csr_plan(adj)
    print(f"[C5] N={N} nnz={nnz} D={D} L={L} hub rows={plan.long_rows.numel()} max deg={int((adj.crow_indices()[1:]-adj.crow_indices()[:-1]).max())}")
    Eu = torch.randn(31668, D, device=dev, requires_grad=True)
    Ei = torch.randn(38048, D, device=dev, requires_grad=True)
    G = torch.randn(N, D, device=dev)

    def fwd():
        return torch.cat(_kernels.lightgcn_propagate(adj, Eu, Ei, L))

    ks, wall = timed(fwd)
    per_layer = 8 * nnz + 4 * (N + 1) + 2 * 4 * N * D + 8 * N * D      # compulsory + fused running sum
    tot = sum(v["avg_us"] * v["count"] for v in ks.values()) / 20
    print(f"  forward: {tot:.1f} us kernel time for {L} layers ({wall*1e6:.0f} us wall) -> {L*per_layer/tot/1e3:.0f} GB/s compulsory, "
          f"{L*nnz/tot/1e3:.2f} G nnz/s; per kernel: " + ", ".join(f"{k} {v['avg_us']:.1f}us x{v['count']//20}" for k, v in ks.items()))

    def fwdbwd():
        Eu.grad = Ei.grad = None
        (fwd() * G).sum().backward()

    ks, wall = timed(fwdbwd)
    tot = sum(v["avg_us"] * v["count"] for v in ks.values()) / 20
    print(f"  fwd+bwd: {tot:.1f} us kernel time in the library ({wall*1e6:.0f} us wall incl. torch glue)")

    # ---- C3: GEMM at CrossNet shapes
    for (M, Nn, K, tA, tB, tag) in [(4096, 352, 352, False, True, "x W^T (DCNHead fwd)"), (352, 352, 4096, True, False, "dlin^T x (dW)"),
                                    (4096, 352, 256, False, False, "H2g U (mix out)"), (4096, 400, 416, False, True, "MLP layer-1 shape"),
                                    (8192, 8192, 8192, False, True, "large square")]:
        A = torch.randn((K, M) if tA else (M, K), device=dev)
        B = torch.randn((Nn, K) if tB else (K, Nn), device=dev)
        C = torch.empty(M, Nn, device=dev)
        ks, _ = timed(lambda: _kernels.gemm(A, B, C, M, Nn, K, A.shape[1], B.shape[1], Nn, tA, tB), reps=10)
        us = ks["gemm_f32"]["avg_us"]
        print(f"[GEMM] {tag:24s} M={M} N={Nn} K={K}: {us:8.1f} us  {2*M*Nn*K/us/1e6:6.1f} TFLOP/s (fp32 MFMA peak 157)")
        Ar = (A.t() if tA else A).contiguous(); Br = (B.t() if tB else B).contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            torch.matmul(Ar, Br)
        torch.cuda.synchronize()
        print(f"        torch.matmul (rocBLAS/hipBLASLt) same shape: {(time.perf_counter()-t0)/10*1e6:8.1f} us wall")

    # ---- C3: DCN-Mix head fwd+bwd at the Avazu shape
    M, d, E, r, Ly = 4096, 352, 4, 64, 3
    head = DCN_MixHead(E, Ly, r, d).to(dev)
    x = (torch.randn(M, d, device=dev) * 0.05).requires_grad_(True)
    Gh = torch.randn(M, d, device=dev)

    def mix():
        head.zero_grad(); x.grad = None
        (head(x) * Gh).sum().backward()

    ks, wall = timed(mix, reps=10)
    tot = sum(v["avg_us"] * v["count"] for v in ks.values()) / 10
    flops = 3 * 2 * M * (d * E * r * 2 + E * r * r) * Ly     # fwd + 2x bwd
    print(f"[C3] DCN_MixHead fwd+bwd M={M} d={d} E={E} r={r} L={Ly}: {tot:.0f} us library kernel time ({wall*1e6:.0f} us wall), "
          f"~{flops/tot/1e6:.1f} TFLOP/s; product launches/step="
          f"{sum(v['count'] for k, v in ks.items() if 'gemm' in k or 'expert' in k) // 10}")
    head2 = DCNHead(3, d).to(dev)

    def v2():
        head2.zero_grad(); x.grad = None
        (head2(x) * Gh).sum().backward()

    ks, wall = timed(v2, reps=10)
    tot = sum(v["avg_us"] * v["count"] for v in ks.values()) / 10
    print(f"[C3] DCNHead fwd+bwd L=3: {tot:.0f} us library kernel time ({wall*1e6:.0f} us wall), {3*3*2*M*d*d/tot/1e6:.1f} TFLOP/s")

    # ---- compressed gathers at B=4096, F=22 (Avazu-like N~2.02M)
    Nv, F, B = 2018025, 22, 4096
    idx = torch.randint(0, Nv, (B, F), device=dev)
    for name, cfg in [("qr div2 mult", {"name": "qr", "divider": 2, "operation": "mult"}), ("cerp 8000", {"name": "cerp", "bucket_size": 8000}),
                      ("vanilla", {"name": "vanilla"})]:
        emb = pkg.get_embedding(cfg, Nv, 16, field_name="x").to(dev)
        Go = torch.randn(B, F, 16, device=dev)

        def fb():
            emb.zero_grad()
            (emb(idx) * Go).sum().backward()

        ks, wall = timed(fb, reps=20)
        print(f"[emb] {name:14s} B={B} F={F}: " + ", ".join(f"{k} {v['avg_us']:.1f}us" for k, v in ks.items()))


if __name__ == "__main__":
    main()
