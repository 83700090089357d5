"""mi_gemm_f32 vs torch (hipBLASLt/rocBLAS) kernel time on the MLP-tail shapes, both measured by
dispatch timestamps (library ring for ours, torch profiler-free event pairs x many for torch)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _kernels
from recsys_benchmark_amd.profiling import KernelTimer
dev = torch.device("cuda")
for (M, N, K, tA, tB, tag) in [(4096, 400, 416, 0, 1, "fwd L1  x W^T"), (4096, 400, 400, 0, 1, "fwd L2/3"),
                               (4096, 416, 400, 0, 0, "dx L1  g W"), (4096, 400, 400, 0, 0, "dx L2/3"),
                               (400, 416, 4096, 1, 0, "dW L1  g^T x"), (400, 400, 4096, 1, 0, "dW L2/3")]:
    A = torch.randn((K, M) if tA else (M, K), device=dev); B = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    f = lambda: _kernels.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, bool(tA), bool(tB))
    for _ in range(3): f()
    torch.cuda.synchronize()
    with KernelTimer(256) as kt:
        for _ in range(20): f()
        torch.cuda.synchronize()
    mine = kt.summary()["gemm_f32"]["avg_us"]
    Ar = A.t() if tA else A; Br = B.t() if tB else B
    for _ in range(3): torch.matmul(Ar, Br)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): torch.matmul(Ar, Br)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    blas = (time.perf_counter() - t0) / 200 * 1e6
    print(f"{tag:14s} M={M} N={N} K={K}: mine {mine:6.1f} us ({2*M*N*K/mine/1e6:5.1f} TF)   torch.matmul in-graph {blas:6.1f} us ({2*M*N*K/blas/1e6:5.1f} TF)")
