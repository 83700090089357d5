"""Do two products of the MLP tail's backward run side by side on one MI355X?  For each layer the weight gradient
(dW = dz^T a) and the input gradient (da = dz W) both depend only on dz, so a second stream can carry the weight
gradients while the first walks the chain.  Measured inside a replayed hipGraph (what a step pays): the pair back to
back on one stream, the pair forked onto two streams, and the 3-layer chain with all weight gradients on the side
stream — for torch.matmul (hipBLASLt / rocBLAS) and for the own fused kernels (csrc/tail.hip)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_benchmark_amd import _lib  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()
P = lambda t: None if t is None else t.data_ptr()   # noqa: E731


def graph_us(fn, n=10, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def main():
    M, N, K = 4096, 400, 416
    torch.manual_seed(0)
    DZ = torch.randn(M, N, device=dev)
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / math.sqrt(K)
    DA, DW = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
    s2 = torch.cuda.Stream()

    def lib_dgrad():
        torch.matmul(DZ, W, out=DA)

    def lib_wgrad():
        torch.matmul(DZ.t(), A, out=DW)

    # own kernels, plain operands (no element work) and with it
    Zl = torch.randn(M, N, device=dev)
    c = [torch.randn(N, device=dev) * 0.3 for _ in range(4)]
    slab = torch.empty(int(lib.mi_tail_wgrad_splits(M, N, K)) * N * K, device=dev)

    def own_dgrad():
        _lib.check(lib.mi_tail_dgrad_gemm(P(DZ), P(Zl), N, P(c[0]), P(c[1]), P(c[2]), P(c[3]), P(W), K, None, 0, None, None, None,
                                          0.0, None, P(DA), K, None, None, M, N, K, _lib.stream_ptr(dev)), "dgrad")

    def own_wgrad():
        _lib.check(lib.mi_tail_wgrad_gemm(P(DZ), P(Zl), N, P(c[0]), P(c[1]), P(c[2]), P(c[3]), P(A), K, None, None, None, 0.0, None,
                                          P(slab), P(DW), M, N, K, _lib.stream_ptr(dev)), "wgrad")

    def forked(main_fn, side_fn):
        def f():
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                side_fn()
            main_fn()
            cur.wait_stream(s2)
        return f

    def chain(main_fn, side_fn, layers=3):
        """side_fn x layers on the side stream, joined once at the end; main_fn x layers on the main stream"""
        def f():
            cur = torch.cuda.current_stream()
            for _ in range(layers):
                s2.wait_stream(cur)          # dz of this layer is ready
                with torch.cuda.stream(s2):
                    side_fn()
                main_fn()
            cur.wait_stream(s2)
        return f

    for name, dg, wg in (("library", lib_dgrad, lib_wgrad), ("own", own_dgrad, own_wgrad)):
        t_d, t_w = graph_us(dg), graph_us(wg)
        t_seq = graph_us(lambda: (dg(), wg()))
        t_fork = graph_us(forked(dg, wg))
        t_chain_seq = graph_us(lambda: [(dg(), wg()) for _ in range(3)], n=4)
        t_chain = graph_us(chain(dg, wg), n=4)
        print(f"{name:8s} dgrad {t_d:6.2f}  wgrad {t_w:6.2f}  back to back {t_seq:6.2f}  forked {t_fork:6.2f} us"
              f"   | 3 layers: one stream {t_chain_seq:6.2f}  weight gradients on a side stream {t_chain:6.2f} us", flush=True)
    # the same fork in eager mode (two real streams), wall per pair
    for name, dg, wg in (("library", lib_dgrad, lib_wgrad),):
        f = forked(dg, wg)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:8s} eager forked pair {e0.elapsed_time(e1) * 1e3 / 200:6.2f} us", flush=True)


if __name__ == "__main__":
    main()
