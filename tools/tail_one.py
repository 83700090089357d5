"""Run ONE tail kernel a few dozen times (for rocprofv3 --pmc passes): python tools/tail_one.py fwd|fwdact|dgrad|dgradmid|wgrad|wgradact"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_benchmark_amd import _lib
dev = torch.device("cuda"); lib = _lib.load()
P = lambda t: None if t is None else t.data_ptr()
S = lambda: _lib.stream_ptr(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
M, N, K, p = 4096, 400, 416, 0.5
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 20; Z = torch.empty(M, N, device=dev)
mu, sc, be = torch.randn(K, device=dev) * .1, torch.rand(K, device=dev) + .5, torch.randn(K, device=dev) * .1
seed = torch.tensor([1], dtype=torch.int64, device=dev)
DY, Zl = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev)
mu_l, al, bz, de = (torch.randn(N, device=dev) * .3 for _ in range(4))
OUT = torch.empty(M, K, device=dev); part = torch.empty(int(lib.mi_tail_part_elems(M, max(N, K))), device=dev)
slab = torch.empty(int(lib.mi_tail_wgrad_splits(M, N, K)) * N * K, device=dev); dW = torch.empty(N, K, device=dev)
fns = {
    "fwd": lambda: lib.mi_tail_fwd_gemm(P(X), K, None, None, None, 0.0, None, 0, P(W), K, P(Z), N, P(part), M, N, K, S()),
    "fwdact": lambda: lib.mi_tail_fwd_gemm(P(X), K, P(mu), P(sc), P(be), p, P(seed), 3, P(W), K, P(Z), N, P(part), M, N, K, S()),
    "dgrad": lambda: lib.mi_tail_dgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(W), K, None, 0, None, None, None, 0.0, None, 0, P(OUT), K, None, M, N, K, S()),
    "dgradmid": lambda: lib.mi_tail_dgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(W), K, P(X), K, P(mu), P(sc), P(be), p, P(seed), 3, P(OUT), K, P(part), M, N, K, S()),
    "wgrad": lambda: lib.mi_tail_wgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(X), K, None, None, None, 0.0, None, 0, P(slab), P(dW), M, N, K, S()),
    "wgradact": lambda: lib.mi_tail_wgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(X), K, P(mu), P(sc), P(be), p, P(seed), 3, P(slab), P(dW), M, N, K, S()),
}
for _ in range(30):
    assert fns[which]() == 0
torch.cuda.synchronize()
