"""Weight-gradient shapes of the MLP tail on mi_gemm_f32: time vs split-K factor (dispatch events).
    python tools/splitk_sweep.py [M N K]      # C[M,N] = A^T[M,K] B[K,N], A stored [K,M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _kernels
from recsys_benchmark_amd.profiling import KernelTimer
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (400, 416, 4096)))
dev = torch.device("cuda")
A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
C = torch.zeros(M, N, device=dev)
ref = A.double().t() @ B.double()
for sk in (0, 1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 16, 21, 32):
    for _ in range(3):
        C.zero_()
        _kernels.gemm(A, B, C, M, N, K, M, N, N, transA=True, splitk=sk, epi="accum" if sk > 1 else "none")
    torch.cuda.synchronize()
    with KernelTimer(64) as kt:
        for _ in range(10):
            C.zero_()
            _kernels.gemm(A, B, C, M, N, K, M, N, N, transA=True, splitk=sk, epi="accum" if sk > 1 else "none")
        torch.cuda.synchronize()
    us = [u for n, u in kt.records if n == "gemm_f32"]
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    print(f"splitk={sk:3d}: {sum(us)/len(us):7.2f} us  ({2*M*N*K/ (sum(us)/len(us))/1e6:6.1f} TFLOP/s)  rel err {err:.1e}")
