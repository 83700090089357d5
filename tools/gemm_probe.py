"""One shape of mi_gemm_f32, a few launches (for rocprofv3 --pmc runs):  gemm_probe.py M N K [transA transB]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _kernels
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 400, 416)))
tA, tB = (int(v) for v in (sys.argv[4:6] if len(sys.argv) > 5 else (0, 1)))
dev = torch.device("cuda")
A = torch.randn((K, M) if tA else (M, K), device=dev)
B = torch.randn((N, K) if tB else (K, N), device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(5):
    _kernels.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, transA=bool(tA), transB=bool(tB))
torch.cuda.synchronize()
print("done")
