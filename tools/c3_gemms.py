"""In-graph time of each product of one DCN-Mix layer at the C3 shape (M=4096, d=352, E=4, r=64), own kernel vs torch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recsys_benchmark_amd._kernels import gemm, gemm_multi, gemm_panel  # noqa: E402

dev = torch.device("cuda")


def graph_us(fn, n=20, reps=20):
    fn()
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
    M, d, E, r = 4096, 352, 4, 64
    Er = E * r
    R = lambda *s: torch.randn(*s, device=dev)          # noqa: E731
    xl, x0, g, gate = R(M, d), R(M, d), R(M, d), R(M, E)
    V, C, U, b, G = R(E, d, r), R(E, r, r), R(E, r, d), R(1, d), R(E, d)
    H1, H2, H2g, out, T = R(M, Er), R(M, Er), R(M, Er), R(M, d), R(M, d)
    dT, dH2g, dZ2, dZ1, gn, dgate = R(M, d), R(M, Er), R(M, Er), R(M, Er), R(M, d), R(M, E)
    cases = [
        ("V   [M,256]=x V_e, tanh", 2.0 * M * d * Er, lambda: gemm(xl, V, H1, M, r, d, d, r, Er, batch=E, sB=d * r, sC=r, epi="tanh"),
         lambda: torch.tanh(torch.matmul(xl, V.permute(1, 0, 2).reshape(d, Er)))),
        ("C   batch 4 x [M,64,64], tanh*gate", 2.0 * M * r * r * E,
         lambda: gemm(H1, C, H2, M, r, r, Er, r, Er, batch=E, sA=r, sB=r * r, sC=r, epi="tanh_gate", rowscale=gate, nrs=E, C2=H2g, ldc2=Er, sC2=r),
         lambda: torch.bmm(H1.view(M, E, r).transpose(0, 1), C)),
        ("U   [M,352]=H2g U, cross epilogue", 2.0 * M * Er * d,
         lambda: gemm(H2g, U, out, M, d, Er, Er, d, d, epi="cross", bias=b, rowscale=gate, nrs=E, R1=xl, ldr1=d, R2=x0, ldr2=d, C2=T, ldc2=d),
         lambda: torch.matmul(H2g, U.view(Er, d))),
        ("dH2g [M,256]=dT U^T", 2.0 * M * Er * d, lambda: gemm(dT, U, dH2g, M, Er, d, d, d, Er, transB=True),
         lambda: torch.matmul(dT, U.view(Er, d).t())),
        ("dZ1 batch 4 x [M,64,64], *tanh'", 2.0 * M * r * r * E,
         lambda: gemm(dZ2, C, dZ1, M, r, r, Er, r, Er, transB=True, batch=E, sA=r, sB=r * r, sC=r, epi="mul_dtanh", R1=H1, ldr1=Er, sR1=r),
         lambda: torch.bmm(dZ2.view(M, E, r).transpose(0, 1), C.transpose(1, 2))),
        ("gn  [M,352]=dZ1 V^T (4 k-groups), add+rank", 2.0 * M * Er * d,
         lambda: gemm(dZ1, V, gn, M, d, r, Er, r, d, transB=True, kgroups=E, gA=r, gB=d * r, epi="add", R1=g, ldr1=d, rowscale=dgate, nrs=E, bias=G),
         lambda: torch.matmul(dZ1, V.permute(0, 2, 1).reshape(Er, d))),
    ]
    panel = {
        "V": lambda: gemm_panel(xl, d, V, r, 1, H1, Er, M, Er, d, gw=r, gstride=d * r, epi="tanh"),
        "U": lambda: gemm_panel(H2g, Er, U, d, 1, out, d, M, d, Er, epi="cross", bias=b, rowscale=gate, nrs=E, R1=xl, R2=x0, C2=T),
        "dH2g": lambda: gemm_panel(dT, d, U, d, 0, dH2g, Er, M, Er, d),
        "gn": lambda: gemm_panel(dZ1, Er, V, r, 0, gn, d, M, d, Er, gw=r, gstride=d * r, epi="add", R1=g, rowscale=dgate, nrs=E, bias=G),
    }
    for name, fl, own, lib in cases:
        t_own, t_lib = graph_us(own), graph_us(lib)
        pk = panel.get(name.split()[0])
        t_p = graph_us(pk) if pk else float("nan")
        print(f"{name:44s} 64x64 tiles {t_own:6.2f} us ({fl / t_own / 1e6:5.1f} TFLOP/s)   panels {t_p:6.2f} us ({fl / t_p / 1e6:5.1f})   "
              f"torch, product only {t_lib:6.2f} us", flush=True)
    flat = torch.zeros(Er * d + E * r * r + E * d * r + E * d, device=dev)
    dU, dC, dV, dG = flat[:Er * d].view(E, r, d), flat[Er * d:Er * d + E * r * r].view(E, r, r), \
        flat[Er * d + E * r * r:Er * d + E * r * r + E * d * r].view(E, d, r), flat[-E * d:].view(E, d)
    probs = [dict(A=H2g, B=dT, C=dU, M=Er, N=d, K=M, lda=Er, ldb=d, ldc=d),
             dict(A=H1, B=dZ2, C=dC, M=r, N=r, K=M, lda=Er, ldb=Er, ldc=r, batch=E, sA=r, sB=r, sC=r * r),
             dict(A=xl, B=dZ1, C=dV, M=d, N=r, K=M, lda=d, ldb=Er, ldc=r, batch=E, sB=r, sC=d * r),
             dict(A=dgate, B=xl, C=dG, M=E, N=d, K=M, lda=E, ldb=d, ldc=d)]
    t1 = graph_us(lambda: gemm_multi(probs, transA=True))
    t3 = graph_us(lambda: gemm_multi(probs * 3, transA=True), n=8)
    fl = 2.0 * M * (Er * d + E * r * r + E * d * r + E * d)
    print(f"weight gradients of one layer, one launch: {t1:6.2f} us ({fl / t1 / 1e6:5.1f} TFLOP/s); of three layers: {t3:6.2f} us "
          f"({3 * fl / t3 / 1e6:5.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
