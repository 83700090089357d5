#!/usr/bin/env python3
"""Where a tail product's fixed cost sits: k_tail_fwd (plain operands) in a replayed graph as a function of the reduction
length K, with and without the statistics epilogue.  T(K) = a + b K: a = launch + prologue + epilogue."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_benchmark_amd import _lib
dev = torch.device("cuda"); lib = _lib.load()
P = lambda t: None if t is None else t.data_ptr()
S = lambda: _lib.stream_ptr(dev)

def graph_us(fn, n=20, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)

M, N = 4096, 400
print("k_tail_fwd<plain>, M=4096, N=400, back to back in a graph")
for K in (32, 64, 128, 224, 416, 832):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K)
    Z = torch.empty(M, N, device=dev); part = torch.empty(int(lib.mi_tail_part_elems(M, N)), device=dev)
    t_stats = graph_us(lambda: _lib.check(lib.mi_tail_fwd_gemm(P(X), K, None, None, None, 0.0, None, P(W), K, P(Z), N, P(part), None, M, N, K, S()), "fwd"))
    t_plain = graph_us(lambda: _lib.check(lib.mi_tail_fwd_gemm(P(X), K, None, None, None, 0.0, None, P(W), K, P(Z), N, None, None, M, N, K, S()), "fwd"))
    print(f"  K={K:4d}  with statistics {t_stats:6.2f} us   without {t_plain:6.2f} us   ({2.0*M*N*K/t_stats/1e6:5.1f} TFLOP/s)")
