import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_benchmark_amd import _lib
dev = torch.device("cuda"); lib = _lib.load()
P = lambda t: None if t is None else t.data_ptr()
M, N, K = 4096, 400, 416
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 20; Z = torch.empty(M, N, device=dev)
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
for dbg, what in ((0, "full"), (64, "full, consumers s_setprio 3"), (128, "full, producers s_setprio 3"), (5, "producers idle"), (64 + 5, "producers idle, consumers prio 3")):
    os.environ["MI_TAIL_DBG"] = str(dbg)
    f = lambda: lib.mi_tail_fwd_gemm(P(X), K, None, None, None, 0.0, None, 0, P(W), K, P(Z), N, None, M, N, K, _lib.stream_ptr(dev))
    print(f"dbg={dbg:3d} {what:50s} {graph_us(f):7.2f} us")
