import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _kernels
from recsys_benchmark_amd.profiling import KernelTimer
dev = torch.device("cuda")
M, N, K = 4096, 400, 416
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for sk in (1, 2, 3):
    f = lambda: _kernels.gemm(A, W, C, M, N, K, K, K, N, transB=True, splitk=sk)
    for _ in range(3): f()
    torch.cuda.synchronize()
    with KernelTimer(256) as kt:
        for _ in range(20): f()
        torch.cuda.synchronize()
    print(sk, {k: round(v["avg_us"], 1) for k, v in kt.summary().items()})
