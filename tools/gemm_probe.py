"""One shape of mi_gemm_f32, a few launches (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _kernels
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 400, 416)))
dev = torch.device("cuda")
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for _ in range(5):
    _kernels.gemm(A, W, C, M, N, K, K, K, N, transB=True)
torch.cuda.synchronize()
print("done")
