"""GPU: the fp32 MFMA GEMM (mi_gemm_f32) against torch fp32 matmul on CPU (exact-fp32 MFMA:
only the summation order differs, rtol 1e-5 / atol scaled with K)."""
import pytest
import torch

from conftest import assert_close

from recsys_benchmark_amd import _kernels

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(shape, gen):
    # asymmetric integer-valued data first (exact in fp32) catches any row/col swap of the MFMA maps
    return torch.randint(-3, 4, shape, generator=gen).float()


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 70, 45), (4096, 352, 352), (33, 5, 7), (130, 416, 400), (7, 130, 1)])
@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
def test_layouts_exact_integers(M, N, K, tA, tB):
    gen = torch.Generator().manual_seed(M * 7 + N + K)
    A = _mk((K, M) if tA else (M, K), gen)
    B = _mk((N, K) if tB else (K, N), gen)
    ref = (A.t() if tA else A) @ (B.t() if tB else B)
    C = torch.empty(M, N, device=DEV)
    _kernels.gemm(A.to(DEV), B.to(DEV), C, M, N, K, A.shape[1], B.shape[1], N, tA, tB)
    assert torch.equal(C.cpu(), ref), "integer-valued GEMM must be exact"


def test_random_fp32_and_epilogues():
    gen = torch.Generator().manual_seed(0)
    M, N, K = 200, 96, 80
    A = torch.randn(M, K, generator=gen)
    W = torch.randn(N, K, generator=gen)
    b = torch.randn(N, generator=gen)
    R1 = torch.randn(M, N, generator=gen)
    R2 = torch.randn(M, N, generator=gen)
    acc = A @ W.t()
    d = lambda t: t.to(DEV)
    C = torch.empty(M, N, device=DEV)
    C2 = torch.empty(M, N, device=DEV)
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, transB=True)
    _kernels.gemm(d(A), d(W), C, epi="bias", bias=d(b), **kw)
    assert_close(C, acc + b, 1e-5, 1e-4)
    _kernels.gemm(d(A), d(W), C, epi="tanh", **kw)
    assert_close(C, torch.tanh(acc), 1e-5, 1e-5)
    _kernels.gemm(d(A), d(W), C, epi="cross", bias=d(b), R1=d(R1), ldr1=N, R2=d(R2), ldr2=N, C2=C2, ldc2=N, **kw)
    assert_close(C, R1 + R2 * (acc + b), 1e-5, 1e-4)
    assert_close(C2, acc + b, 1e-5, 1e-4)
    rs = torch.randn(M, 3, generator=gen)
    _kernels.gemm(d(A), d(W), C, epi="cross", bias=d(b), R1=d(R1), ldr1=N, R2=d(R2), ldr2=N, rowscale=d(rs), nrs=3, **kw)
    assert_close(C, R1 + R2 * (acc + b[None] * rs.sum(1, keepdim=True)), 1e-5, 1e-4)
    _kernels.gemm(d(A), d(W), C, epi="add", R1=d(R1), ldr1=N, **kw)
    assert_close(C, R1 + acc, 1e-5, 1e-4)
    H = torch.tanh(R1)
    _kernels.gemm(d(A), d(W), C, epi="mul_dtanh", R1=d(H), ldr1=N, **kw)
    assert_close(C, acc * (1 - H * H), 1e-5, 1e-4)
    C.copy_(d(R1))
    _kernels.gemm(d(A), d(W), C, epi="accum", **kw)
    assert_close(C, R1 + acc, 1e-5, 1e-4)


def test_batched_column_slices_kgroups_and_gate():
    gen = torch.Generator().manual_seed(1)
    M, d_, E, r = 150, 40, 3, 8
    x = torch.randn(M, d_, generator=gen)
    V = torch.randn(E, d_, r, generator=gen)
    gate = torch.randn(M, E, generator=gen)
    dv = lambda t: t.to(DEV)
    # H1[:, e*r:(e+1)*r] = tanh(x @ V_e): batch over experts writing column slices of one [M, E*r] matrix
    H1 = torch.empty(M, E * r, device=DEV)
    H1g = torch.empty(M, E * r, device=DEV)
    _kernels.gemm(dv(x), dv(V), H1, M, r, d_, d_, r, E * r, batch=E, sA=0, sB=d_ * r, sC=r, epi="tanh_gate",
                  rowscale=dv(gate), nrs=E, C2=H1g, ldc2=E * r, sC2=r)
    ref = torch.cat([torch.tanh(x @ V[e]) for e in range(E)], 1)
    assert_close(H1, ref, 1e-5, 1e-5)
    assert_close(H1g, torch.cat([torch.tanh(x @ V[e]) * gate[:, e:e + 1] for e in range(E)], 1), 1e-5, 1e-5)
    # sum over experts inside one launch: out = sum_e H1_e @ V_e^T (K-groups)
    out = torch.empty(M, d_, device=DEV)
    _kernels.gemm(H1, dv(V), out, M, d_, r, E * r, r, d_, transB=True, kgroups=E, gA=r, gB=d_ * r)
    assert_close(out, sum(ref[:, e * r:(e + 1) * r] @ V[e].t() for e in range(E)), 1e-5, 1e-4)
