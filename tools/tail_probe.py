"""Correctness + cost of the fused MLP-tail kernels (csrc/tail.hip) at the headline shapes, against torch on the same
GPU: every kernel vs a float64 evaluation, and wall time per kernel inside a replayed hipGraph (what a step pays),
next to torch.matmul (hipBLASLt / rocBLAS) for the same products."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from recsys_benchmark_amd import _lib
from tail_helpers import tail_keep_scale
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

import ctypes
def make_masks(seed, specs, M):
    """specs: [(salt, p, ld)] -> list of uint8 device tensors (keep bits), one launch"""
    n = len(specs)
    bits = [torch.empty(M * ld // 8, dtype=torch.uint8, device=dev) for _, _, ld in specs]
    salts = (ctypes.c_int64 * n)(*[s_ for s_, _, _ in specs]); ps = (ctypes.c_float * n)(*[p_ for _, p_, _ in specs])
    lds = (ctypes.c_int32 * n)(*[l for _, _, l in specs]); ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bits])
    _lib.check(lib.mi_tail_dropout_masks(P(seed), n, ctypes.addressof(salts), ctypes.addressof(ps), ctypes.addressof(lds),
                                         ctypes.addressof(ptrs), M, S()), "masks")
    return bits

def relerr(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))

def run(M, N, K, p, check=True):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    X = torch.randn(M, K, generator=g).to(dev)          # layer input (plain) / previous z
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    mu, sc, be = (torch.randn(K, generator=g) * 0.1).to(dev), (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.1).to(dev)
    seed = torch.tensor([12345], dtype=torch.int64, device=dev)
    bitsK, bitsN = make_masks(seed, [(3, p, K), (5, p, N)], M)
    t_masks = graph_us(lambda: make_masks(seed, [(3, p, K), (5, p, N), (7, p, N)], M))
    Z = torch.empty(M, N, device=dev)
    part = torch.empty(int(lib.mi_tail_part_elems(M, N)), device=dev)
    # ---- forward, plain input
    def fwd_plain():
        _lib.check(lib.mi_tail_fwd_gemm(P(X), K, None, None, None, 0.0, None, P(W), K, P(Z), N, P(part), None, M, N, K, S()), "fwd")
    def fwd_act():
        _lib.check(lib.mi_tail_fwd_gemm(P(X), K, P(mu), P(sc), P(be), p, P(bitsK), P(W), K, P(Z), N, P(part), None, M, N, K, S()), "fwd")
    out = {}
    if check:
        fwd_plain(); ref = X.double() @ W.double().t()
        out["fwd_plain"] = relerr(Z, ref)
        MT = (M + 63) // 64
        pm = part.view(MT, N, 2)
        # merged stats vs direct
        gm = torch.empty(N, device=dev); 
        mu_o, sc_o, be_o, rs_o = (torch.empty(N, device=dev) for _ in range(4))
        rmean, rvar = torch.zeros(N, device=dev), torch.ones(N, device=dev)
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        gamma, beta = (torch.rand(N, generator=g) + 0.5).to(dev), torch.randn(N, generator=g).to(dev)
        _lib.check(lib.mi_tail_bn_finalize_fwd(P(part), M, N, P(gamma), P(beta), None, P(rmean), P(rvar), 0.1, 1e-5, P(nbt), None,
                                               P(mu_o), P(sc_o), P(be_o), P(rs_o), S()), "fin")
        out["bn_mean"] = relerr(mu_o, ref.mean(0)); out["bn_rstd"] = relerr(rs_o, (ref.var(0, unbiased=False) + 1e-5).rsqrt())
        out["bn_rvar"] = relerr(rvar, 0.9 + 0.1 * ref.var(0, unbiased=True))
        fwd_act()
        keep = tail_keep_scale(12345, 3, M, K, p, device=dev).double()
        A = (((X.double() - mu.double()) * sc.double() + be.double()).clamp_min(0) * keep)
        out["fwd_act"] = relerr(Z, A @ W.double().t())
    t_fwd_plain = graph_us(fwd_plain); t_fwd_act = graph_us(fwd_act)
    Zt = torch.empty(M, N, device=dev)
    t_torch_fwd = graph_us(lambda: torch.matmul(X, W.t(), out=Zt))
    # ---- dgrad: DY [M,N], Zl [M,N] consts over N; W [N,K]; prev = (X as z_prev, mu, sc, be, p)
    DY = torch.randn(M, N, generator=g).to(dev); Zl = torch.randn(M, N, generator=g).to(dev)
    mu_l, al, bz, de = ((torch.randn(N, generator=g) * 0.3).to(dev) for _ in range(4))
    OUT = torch.empty(M, K, device=dev); dpart = torch.empty(int(lib.mi_tail_part_elems(M, K)), device=dev)
    def dgrad_mid():
        _lib.check(lib.mi_tail_dgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(W), K, P(X), K, P(mu), P(sc), P(be), p, P(bitsK),
                                          P(OUT), K, P(dpart), None, M, N, K, S()), "dgrad")
    def dgrad_plain():
        _lib.check(lib.mi_tail_dgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(W), K, None, 0, None, None, None, 0.0, None,
                                          P(OUT), K, None, None, M, N, K, S()), "dgrad")
    if check:
        dz = al.double() * DY.double() + bz.double() * (Zl.double() - mu_l.double()) + de.double()
        da = dz @ W.double()
        dgrad_plain(); out["dgrad_plain"] = relerr(OUT, da)
        dgrad_mid()
        pre = (X.double() - mu.double()) * sc.double() + be.double()
        pre32 = torch.addcmul(be, X - mu, sc)      # the kernel decides the ReLU mask in fp32 (fma)
        dy_prev = da * keep * (pre32 > 0)
        out["dgrad_mid"] = relerr(OUT, dy_prev)
        MT = (M + 63) // 64
        pp = dpart.view(MT, K, 2).double().sum(0)
        out["dgrad_sum_dy"] = relerr(pp[:, 0], dy_prev.sum(0)); out["dgrad_sum_dyz"] = relerr(pp[:, 1], (dy_prev * (X.double() - mu.double())).sum(0))
    t_dgrad_mid = graph_us(dgrad_mid); t_dgrad_plain = graph_us(dgrad_plain)
    t_torch_dgrad = graph_us(lambda: torch.matmul(DY, W, out=OUT))
    # ---- wgrad
    splits = int(lib.mi_tail_wgrad_splits(M, N, K))
    slab = torch.empty(splits * N * K, device=dev); dW = torch.empty(N, K, device=dev)
    def wgrad_act():
        _lib.check(lib.mi_tail_wgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(X), K, P(mu), P(sc), P(be), p, P(bitsK),
                                          P(slab), P(dW), M, N, K, S()), "wgrad")
    def wgrad_plain():
        _lib.check(lib.mi_tail_wgrad_gemm(P(DY), P(Zl), N, P(mu_l), P(al), P(bz), P(de), P(X), K, None, None, None, 0.0, None,
                                          P(slab), P(dW), M, N, K, S()), "wgrad")
    if check:
        wgrad_plain(); out["wgrad_plain"] = relerr(dW, dz.t() @ X.double())
        wgrad_act(); out["wgrad_act"] = relerr(dW, dz.t() @ A)
    t_wgrad_act = graph_us(wgrad_act); t_wgrad_plain = graph_us(wgrad_plain)
    dWt = torch.empty(N, K, device=dev)
    t_torch_wgrad = graph_us(lambda: torch.matmul(DY.t(), X, out=dWt))
    # ---- head
    w4 = torch.randn(N, generator=g).to(dev); b4 = torch.randn(1, generator=g).to(dev); add = torch.randn(M, generator=g).to(dev)
    o = torch.empty(M, device=dev)
    muN, scN, beN = mu_l, (al.abs() + 0.5), bz
    def head_fwd():
        _lib.check(lib.mi_tail_head_fwd(P(Zl), N, P(muN), P(scN), P(beN), p, P(bitsN), P(w4), P(b4), P(add), P(o), M, N, S()), "head")
    nblk = int(lib.mi_tail_head_blocks(M))
    gvec = torch.randn(M, generator=g).to(dev); DYh = torch.empty(M, N, device=dev)
    hpart = torch.empty(nblk * N * 2, device=dev); wpart = torch.empty(nblk * (N + 4), device=dev)
    def head_bwd():
        _lib.check(lib.mi_tail_head_bwd(P(Zl), N, P(muN), P(scN), P(beN), p, P(bitsN), P(gvec), P(w4), P(DYh), P(hpart), P(wpart), M, N, S()), "headb")
    if check:
        keepN = tail_keep_scale(12345, 5, M, N, p, device=dev).double()
        pre32 = torch.addcmul(beN, Zl - muN, scN)
        aN = pre32.double().clamp_min(0) * keepN
        head_fwd(); out["head_fwd"] = relerr(o, aN @ w4.double() + b4.double() + add.double())
        head_bwd()
        dyh = gvec.double()[:, None] * w4.double()[None] * keepN * (pre32 > 0)
        out["head_dy"] = relerr(DYh, dyh)
        hp = hpart.view(nblk, N, 2).double().sum(0)
        out["head_sum_dy"] = relerr(hp[:, 0], dyh.sum(0)); out["head_sum_dyz"] = relerr(hp[:, 1], (dyh * (Zl.double() - muN.double())).sum(0))
        wp = wpart.view(nblk, N + 4).double().sum(0)
        out["head_dw"] = relerr(wp[:N], (gvec.double()[:, None] * aN).sum(0)); out["head_db"] = relerr(wp[N:N+1], gvec.double().sum().view(1))
    t_head_fwd = graph_us(head_fwd); t_head_bwd = graph_us(head_bwd)
    print(f"M={M} N={N} K={K} p={p}")
    for k, v in out.items(): print(f"   {k:16s} max rel err {v:.2e}")
    fl = 2.0 * M * N * K
    for name, t in (("fwd plain", t_fwd_plain), ("fwd act", t_fwd_act), ("torch fwd", t_torch_fwd), ("dgrad mid", t_dgrad_mid), ("dgrad plain", t_dgrad_plain),
                    ("torch dgrad", t_torch_dgrad), ("wgrad act", t_wgrad_act), ("wgrad plain", t_wgrad_plain), ("torch wgrad", t_torch_wgrad)):
        print(f"   {name:12s} {t:7.2f} us/kernel in-graph  {fl / t / 1e6:6.1f} TFLOP/s")
    print(f"   head fwd {t_head_fwd:.2f} us   head bwd {t_head_bwd:.2f} us   dropout masks (3 layers, one launch) {t_masks:.2f} us")

if __name__ == "__main__":
    run(200, 40, 48, 0.5)          # ragged small case: edges everywhere (features % 8 == 0 for the keep bits)
    run(4096, 400, 416, 0.5)
    run(4096, 400, 400, 0.5)
