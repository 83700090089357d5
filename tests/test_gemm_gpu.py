"""GPU: the fp32 MFMA GEMM (mi_gemm_f32) against torch fp32 matmul on CPU (exact-fp32 MFMA:
only the summation order differs, rtol 1e-5 / atol scaled with K)."""
import os

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


@pytest.mark.parametrize("tA,tB", [(True, False), (False, True), (False, False), (True, True)])
def test_multi_problem_launch_equals_one_launch_per_problem(tA, tB):
    """mi_gemm_f32_multi: independent problems of one operand layout in one launch — mixed sizes (ragged tiles, a single
    row), batched column slices, split-K into zero-filled views of ONE flat buffer, accumulate; integer-valued data, so
    every result is exact whatever the split; 19 problems = two launches."""
    gen = torch.Generator().manual_seed(11 + 2 * tA + tB)
    shapes = [(256, 352, 4096), (64, 64, 4096), (352, 64, 1024), (4, 352, 4096), (100, 70, 45), (33, 5, 7), (1, 130, 300)]
    shapes = shapes + shapes[:6] + shapes[1:7]
    probs, refs = [], []
    total = sum(M * N for M, N, _ in shapes) + 2 * 3 * 64 * 48
    flat = torch.zeros(total, device=DEV)
    off = 0
    for i, (M, N, K) in enumerate(shapes):
        A = _mk((K, M) if tA else (M, K), gen)
        B = _mk((N, K) if tB else (K, N), gen)
        C = flat[off:off + M * N].view(M, N)
        off += M * N
        ref = (A.t() if tA else A) @ (B.t() if tB else B)
        acc = i % 5 == 4
        if acc:
            C.copy_(_mk((M, N), gen))
            ref = ref + C.cpu()
        probs.append(dict(A=A.to(DEV), B=B.to(DEV), C=C, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N,
                          accumulate=acc, splitk=0 if i % 3 else 1))
        refs.append(ref)
    # batched: 3 slices of wide operands (weight-gradient form only: column slices of [K, 3*64] and [K, 3*48])
    if tA and not tB:
        for _ in range(2):
            A, B = _mk((512, 3 * 64), gen), _mk((512, 3 * 48), gen)
            C = flat[off:off + 3 * 64 * 48].view(3, 64, 48)
            off += 3 * 64 * 48
            probs.append(dict(A=A.to(DEV), B=B.to(DEV), C=C, M=64, N=48, K=512, lda=192, ldb=144, ldc=48, batch=3, sA=64,
                              sB=48, sC=64 * 48))
            refs.append(torch.stack([A[:, 64 * e:64 * e + 64].t() @ B[:, 48 * e:48 * e + 48] for e in range(3)]))
    _kernels.gemm_multi(probs, transA=tA, transB=tB)
    for q, ref in zip(probs, refs):
        assert torch.equal(q["C"].cpu(), ref), (q["M"], q["N"], q["K"])
    assert torch.equal(flat[off:], torch.zeros_like(flat[off:])), "nothing written past the last problem"


def test_multi_problem_launch_argument_checks():
    from recsys_benchmark_amd import _lib

    z = torch.zeros(64, 64, device=DEV)
    ok = dict(A=z, B=z, C=z.clone(), M=64, N=64, K=64, lda=64, ldb=64, ldc=64)
    _kernels.gemm_multi([], transA=True)                                # nothing to do
    _kernels.gemm_multi([dict(ok, M=0)], transA=True)                   # empty problem: skipped
    with pytest.raises(_lib.MI355XLibraryError):
        _kernels.gemm_multi([dict(ok, K=-1)], transA=True)
    with pytest.raises(_lib.MI355XLibraryError):
        _kernels.gemm_multi([dict(ok, splitk=-2)], transA=True)


def test_add_epilogue_with_rank_term():
    """epi 'add' + (rowscale, bias): C = R1 + A.B (+ R2) + rowscale[M,E] . bias[E,N] — the gate's share of a DCN-Mix
    layer's input gradient folded into the product that precedes it."""
    gen = torch.Generator().manual_seed(21)
    for M, N, K, E in [(200, 96, 80, 4), (4096, 352, 64, 4), (70, 20, 33, 1), (65, 130, 40, 8)]:
        A, W = _mk((M, K), gen), _mk((N, K), gen)
        R1, R2 = _mk((M, N), gen), _mk((M, N), gen)
        rs, G = _mk((M, E), gen), _mk((E, N), gen)
        d = lambda t: t.to(DEV)                                                    # noqa: E731
        C = torch.empty(M, N, device=DEV)
        kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, transB=True)
        _kernels.gemm(d(A), d(W), C, epi="add", R1=d(R1), ldr1=N, R2=d(R2), ldr2=N, rowscale=d(rs), nrs=E, bias=d(G), **kw)
        assert torch.equal(C.cpu(), R1 + A @ W.t() + R2 + rs @ G)
        _kernels.gemm(d(A), d(W), C, epi="add", R1=d(R1), ldr1=N, rowscale=d(rs), nrs=E, bias=d(G), **kw)
        assert torch.equal(C.cpu(), R1 + A @ W.t() + rs @ G)


@pytest.mark.parametrize("M,N,K", [(4096, 352, 256), (4096, 256, 352), (200, 40, 48), (67, 8, 4), (1000, 416, 416), (130, 460, 64)])
@pytest.mark.parametrize("layout", [0, 1])
def test_panel_gemm_layouts_groups_and_epilogues(M, N, K, layout):
    """mi_gemm_f32_panel (64-row panels x column ranges, one workgroup per CU) against torch on integer-valued data (exact):
    both B layouts, ungrouped and grouped along B's contiguous side (the experts), every epilogue it has."""
    gen = torch.Generator().manual_seed(M + 3 * N + 7 * K + layout)
    d = lambda t: t.to(DEV)                                                         # noqa: E731
    A = _mk((M, K), gen)
    # ungrouped
    Bm = _mk((N, K) if layout == 0 else (K, N), gen)
    prod = A @ (Bm.t() if layout == 0 else Bm)
    C = torch.empty(M, N, device=DEV)
    assert _kernels.gemm_panel(d(A), K, d(Bm), Bm.shape[1], layout, C, N, M, N, K)
    assert torch.equal(C.cpu(), prod)
    b, R1, R2, rs = _mk((N,), gen), _mk((M, N), gen), _mk((M, N), gen), _mk((M, 3), gen)
    C2 = torch.empty(M, N, device=DEV)
    assert _kernels.gemm_panel(d(A), K, d(Bm), Bm.shape[1], layout, C, N, M, N, K, epi="cross", bias=d(b), R1=d(R1), R2=d(R2),
                               rowscale=d(rs), nrs=3, C2=C2)
    lin = prod + b[None] * rs.sum(1, keepdim=True)
    assert torch.equal(C2.cpu(), lin) and torch.equal(C.cpu(), R1 + R2 * lin)
    assert _kernels.gemm_panel(d(A), K, d(Bm), Bm.shape[1], layout, C, N, M, N, K, epi="cross", bias=d(b), R1=d(R1), R2=d(R2))
    assert torch.equal(C.cpu(), R1 + R2 * (prod + b))
    G = _mk((3, N), gen)
    assert _kernels.gemm_panel(d(A), K, d(Bm), Bm.shape[1], layout, C, N, M, N, K, epi="add", R1=d(R1), R2=d(R2), rowscale=d(rs),
                               nrs=3, bias=d(G))
    assert torch.equal(C.cpu(), R1 + prod + R2 + rs @ G)
    assert _kernels.gemm_panel(d(A), K, d(Bm), Bm.shape[1], layout, C, N, M, N, K, epi="add", R1=d(R1))
    assert torch.equal(C.cpu(), R1 + prod)
    assert _kernels.gemm_panel(d(A), K, d(Bm * 0.25), Bm.shape[1], layout, C, N, M, N, K, epi="tanh")
    assert_close(C, torch.tanh(prod * 0.25), 1e-6, 1e-6)
    # grouped: the contiguous side of B in groups of gw (K for layout 0, N for layout 1), groups gstride apart
    side = K if layout == 0 else N
    for gw in (4, 8, 16, 64):
        if side % gw or side // gw < 2:
            continue
        ng, other = side // gw, (N if layout == 0 else K)
        Bg = _mk((ng, other, gw), gen)                         # group e: [other, gw], like V[e] = [d, r]
        full = Bg.permute(1, 0, 2).reshape(other, side)        # [other, side]: row o = the groups' rows side by side
        ref = A @ (full.t() if layout == 0 else full)
        assert _kernels.gemm_panel(d(A), K, d(Bg), gw, layout, C, N, M, N, K, gw=gw, gstride=other * gw)
        assert torch.equal(C.cpu(), ref), gw


def test_panel_gemm_declines_what_it_does_not_cover():
    A, B, C = torch.zeros(8, 6, device=DEV), torch.zeros(8, 6, device=DEV), torch.zeros(8, 8, device=DEV)
    assert not _kernels.gemm_panel(A, 6, B, 6, 0, C, 8, 8, 8, 6)                   # K % 4
    assert not _kernels.gemm_panel(A, 6, B, 6, 0, C, 8, 8, 8, 4, epi="mul_dtanh")   # an epilogue it does not have
    assert not _kernels.gemm_panel(A[:, 1:], 6, B, 6, 0, C, 8, 8, 8, 4)             # misaligned operand


def test_multi_problem_launch_reduction_major_form():
    """The weight-gradient form A[k][m]^T B[k][n] (both operands reduction-major) in one multi-problem launch: ragged tiles
    (100 x 68), a 4-row problem, K tails (45, 1000), one k-tile, batched column slices, accumulate, explicit and automatic
    K-slices; integer-valued data: exact."""
    gen = torch.Generator().manual_seed(77)
    shapes = [(400, 416, 4096), (100, 68, 45), (4, 352, 4096), (64, 64, 32), (256, 352, 1000), (352, 64, 4096), (8, 4, 7)]
    probs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        A, B = _mk((K, M), gen), _mk((K, N), gen)
        acc = i % 3 == 2
        C = _mk((M, N), gen).to(DEV) if acc else torch.zeros(M, N, device=DEV)
        ref = A.t() @ B + (C.cpu() if acc else 0)
        probs.append(dict(A=A.to(DEV), B=B.to(DEV), C=C, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, accumulate=acc,
                          splitk=(0, 1, 3)[i % 3]))
        refs.append(ref)
    A, B = _mk((512, 3 * 64), gen), _mk((512, 3 * 48), gen)
    C = torch.zeros(3, 64, 48, device=DEV)
    probs.append(dict(A=A.to(DEV), B=B.to(DEV), C=C, M=64, N=48, K=512, lda=192, ldb=144, ldc=48, batch=3, sA=64, sB=48, sC=64 * 48))
    refs.append(torch.stack([A[:, 64 * e:64 * e + 64].t() @ B[:, 48 * e:48 * e + 48] for e in range(3)]))
    _kernels.gemm_multi(probs, transA=True)
    for q, ref in zip(probs, refs):
        assert torch.equal(q["C"].cpu(), ref), (q["M"], q["N"], q["K"])


def test_multi_problem_launch_reduction_major_lds_dma_kernel():
    """A launch whose problems ALL fit the LDS-DMA kernels (k_gemm_tn_multi: K % 32 == 0, M % 4 == N % 4 == 0, aligned) must
    run there (checked through the launch's name in the dispatch-event ring) and be exact on integer-valued data: the C2
    tail's shapes (400 x 416: the 64-row tiles' last 32-row block is empty, the one before half full), problems of 1, 2, 3
    k-tiles, a 4 x 4 corner, batched column slices with strides, accumulate, explicit K-slices (also more slices than k-tiles: empty slices) and the library's cut."""
    from recsys_benchmark_amd.profiling import KernelTimer

    gen = torch.Generator().manual_seed(1234)
    shapes = [(400, 416, 4096), (400, 400, 4096), (4, 352, 4096), (64, 64, 32), (100, 68, 64), (36, 8, 96), (352, 64, 4096),
              (8, 4, 160), (4, 4, 32), (132, 260, 2048)]
    probs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        A, B = _mk((K, M), gen), _mk((K, N), gen)
        acc = i % 3 == 2
        C = _mk((M, N), gen).to(DEV) if acc else torch.zeros(M, N, device=DEV)
        ref = A.t() @ B + (C.cpu() if acc else 0)
        probs.append(dict(A=A.to(DEV), B=B.to(DEV), C=C, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, accumulate=acc,
                          splitk=(0, 1, 3, 7)[i % 4]))
        refs.append(ref)
    A, B = _mk((512, 3 * 64), gen), _mk((512, 3 * 48), gen)
    C = torch.zeros(3, 64, 48, device=DEV)
    probs.append(dict(A=A.to(DEV), B=B.to(DEV), C=C, M=64, N=48, K=512, lda=192, ldb=144, ldc=48, batch=3, sA=64, sB=48, sC=64 * 48))
    refs.append(torch.stack([A[:, 64 * e:64 * e + 64].t() @ B[:, 48 * e:48 * e + 48] for e in range(3)]))
    with KernelTimer(64) as kt:
        _kernels.gemm_multi(probs, transA=True)
    names = [k for k, _ in kt.records]
    assert names == ["gemm_f32_multi" if os.environ.get("MI_GEMM_TN_DMA") == "0" else "gemm_tn_multi"], names
    for q, ref in zip(probs, refs):
        assert torch.equal(q["C"].cpu(), ref), (q["M"], q["N"], q["K"])
    # random fp32 at the tail's shape against float64: same bound as any fp32 product
    A, B = torch.randn(4096, 400, generator=gen), torch.randn(4096, 416, generator=gen)
    C = torch.zeros(400, 416, device=DEV)
    _kernels.gemm_multi([dict(A=A.to(DEV), B=B.to(DEV), C=C, M=400, N=416, K=4096, lda=400, ldb=416, ldc=416)], transA=True)
    ref = A.double().t() @ B.double()
    assert_close(C, ref.float(), 1e-5, 1e-3)


def test_multi_problem_launch_carrying_prefetch_riders():
    """mi_gemm_f32_multi_ride: the weight-gradient launch with extra workgroups that touch the table rows of the next batch's
    lookup (DeepFM.prefetch_next).  The products are unchanged bit for bit (integer data, both kernels), out-of-range ids among
    the riders' are skipped, a launch that cannot carry the job runs it as a launch of its own, and a job without products is
    just that job."""
    from recsys_benchmark_amd import _lib
    from recsys_benchmark_amd.profiling import KernelTimer

    gen = torch.Generator().manual_seed(77)
    N, D, B, F = 5000, 16, 300, 7
    table = torch.randn(N, 32, device=DEV)
    ids = torch.randint(0, N // F, (B, F), generator=gen)
    ids[3, 2] = 10 ** 9                      # beyond the table: skipped
    ids[5, 0] = -4
    ids = ids.to(DEV)
    off = (torch.arange(F) * (N // F)).to(DEV)
    job = _kernels.PrefetchRowsJob(ids.data_ptr(), off.data_ptr(), table.data_ptr(), table.data_ptr() + 4 * D, 32, 32, B, N, F)
    for shapes, kernel in (([(400, 416, 4096), (64, 64, 32), (132, 260, 2048)], "gemm_tn_multi"), ([(30, 50, 70)], "gemm_f32_multi")):
        probs, refs = [], []
        for (M, Nn, K) in shapes:
            A, Bm = _mk((K, M), gen), _mk((K, Nn), gen)
            probs.append(dict(A=A.to(DEV), B=Bm.to(DEV), C=torch.zeros(M, Nn, device=DEV), M=M, N=Nn, K=K, lda=M, ldb=Nn, ldc=Nn))
            refs.append(A.t() @ Bm)
        with KernelTimer(16) as kt:
            _kernels.gemm_multi(probs, transA=True, ride=job)
        names = [k for k, _ in kt.records]
        if os.environ.get("MI_GEMM_TN_DMA") != "0":
            assert names == ([kernel] if kernel == "gemm_tn_multi" else ["prefetch_rows", kernel]), names
        for q, ref in zip(probs, refs):
            assert torch.equal(q["C"].cpu(), ref)
    with KernelTimer(16) as kt:
        _kernels.gemm_multi([], transA=True, ride=job)
    assert [k for k, _ in kt.records] == ["prefetch_rows"]
    torch.cuda.synchronize()
    _lib.check_index_errors()
