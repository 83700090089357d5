"""GPU parity of the CrossNet heads and the DCN-Mix / DCNv2 models (fp32 MFMA GEMMs + fused
epilogues) vs the reference goldens and the oracle.  Exact-fp32 MFMA: only summation order and
tanhf-vs-CPU-tanh differ; activations rtol 2e-5 / atol 2e-6, gradients rtol 2e-4 / atol 2e-5."""
import pytest
import torch

from conftest import assert_close, golden_names, load_golden
from oracle import reference_ops as ro

import recsys_benchmark_amd as pkg
from recsys_benchmark_amd.layer_dcn import DCN_MixHead, DCNHead

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[True, False], ids=["own-tail", "library-tail"])
def _both_tails(request, monkeypatch):
    """Every test of this file runs with the MLP tail on the own fused kernels (the default) and on the general path
    (library products + the fused BatchNorm passes), so both stay covered whatever the default is."""
    from recsys_benchmark_amd import mlp as _mlp_mod

    monkeypatch.setattr(_mlp_mod, "FUSED_TAIL", request.param)
DEV = "cuda"


def test_dcn_head_golden():
    g = load_golden("dcn_head")
    head = DCNHead(3, 24)
    head.load_state_dict(g.group("param/"), strict=True)
    head.to(DEV)
    x = g.t("x").to(DEV).requires_grad_(True)
    out = head(x)
    assert_close(out, g.t("out"), 2e-5, 2e-6, "out")
    (out * g.t("G").to(DEV)).sum().backward()
    assert_close(x.grad, g.t("grad_x"), 2e-4, 2e-5, "grad x")
    named = dict(head.named_parameters())
    for k, ref in g.group("grad/").items():
        assert_close(named[k].grad, ref, 2e-4, 2e-5, k)


def test_dcn_mixhead_golden():
    g = load_golden("dcn_mixhead")
    head = DCN_MixHead(num_experts=4, num_layers=3, rank=8, hidden_size=24)
    head.load_state_dict(g.group("param/"), strict=True)
    head.to(DEV)
    x = g.t("x").to(DEV).requires_grad_(True)
    out = head(x * 0.5)
    assert_close(out, g.t("out"), 2e-5, 2e-6, "out")
    (out * g.t("G").to(DEV)).sum().backward()
    assert_close(x.grad, g.t("grad_x"), 2e-4, 2e-5, "grad x")
    named = dict(head.named_parameters())
    for k, ref in g.group("grad/").items():
        assert_close(named[k].grad, ref, 2e-4, 2e-5, k)


@pytest.mark.parametrize("M,d,L", [(4096, 352, 3), (100, 416, 2), (33, 20, 1), (7, 64, 0)])
def test_dcn_head_vs_oracle(M, d, L):
    gen = torch.Generator().manual_seed(M + d)
    head = DCNHead(L, d)
    X = torch.randn(M, d, generator=gen) * 0.3
    G = torch.randn(M, d, generator=gen)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in head.state_dict().items()}
    x = X.clone().requires_grad_(True)
    ref = ro.dcn_head(x, p, L)
    (ref * G).sum().backward()
    head.to(DEV)
    xd = X.to(DEV).requires_grad_(True)
    out = head(xd)
    assert_close(out, ref, 1e-4, 1e-4, "out")
    (out * G.to(DEV)).sum().backward()
    scale = max(1.0, float(x.grad.abs().max()))
    assert_close(xd.grad, x.grad, 1e-3, 1e-4 * scale, "grad x")
    for k, v in head.named_parameters():
        s = max(1.0, float(p[k].grad.abs().max()))
        assert_close(v.grad, p[k].grad, 1e-3, 1e-4 * s, k)


@pytest.mark.parametrize("M,d,E,r,L", [(4096, 352, 4, 64, 3), (50, 40, 3, 8, 2), (9, 16, 1, 4, 1)])
def test_dcn_mixhead_vs_oracle(M, d, E, r, L):
    gen = torch.Generator().manual_seed(M + d + r)
    head = DCN_MixHead(E, L, r, d)
    with torch.no_grad():
        for b in head.biases:
            b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
        # kaiming-normal on [E,d,1] gates has std ~1.4: scale inputs so 3 layers stay O(1)
    X = torch.randn(M, d, generator=gen) * (0.05 if d > 100 else 0.3)
    G = torch.randn(M, d, generator=gen)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in head.state_dict().items()}
    x = X.clone().requires_grad_(True)
    ref = ro.dcn_mix_head(x, p, L)
    (ref * G).sum().backward()
    head.to(DEV)
    xd = X.to(DEV).requires_grad_(True)
    out = head(xd)
    assert_close(out, ref, 2e-4, 1e-4 * max(1.0, float(ref.abs().max())), "out")
    (out * G.to(DEV)).sum().backward()
    assert_close(xd.grad, x.grad, 2e-3, 2e-4 * max(1.0, float(x.grad.abs().max())), "grad x")
    for k, v in head.named_parameters():
        s = max(1.0, float(p[k].grad.abs().max()))
        assert_close(v.grad, p[k].grad, 2e-3, 2e-4 * s, k)


@pytest.mark.parametrize("name", golden_names("dcn_mix_") + golden_names("dcnv2_"))
def test_dcn_models_match_reference_golden(name):
    g = load_golden(name)
    p = g.group("param/")
    dims = g["field_dims"].tolist()
    if name.startswith("dcn_mix_vanilla"):
        cfg = {"name": "dcn_mix", "num_factor": 4, "hidden_sizes": [8, 8], "num_layers": 2, "num_experts": 3, "rank": 5,
               "p_dropout": 0.0, "compile_model": False}
    elif name.startswith("dcn_mix_qr"):
        cfg = {"name": "dcn_mix", "num_factor": 4, "hidden_sizes": [8], "num_layers": 3, "num_experts": 4, "rank": 6,
               "p_dropout": 0.0, "compile_model": False, "embedding_config": {"name": "qr", "divider": 2, "operation": "mult"}}
    elif name.startswith("dcnv2_stacked"):
        cfg = {"name": "dcn", "num_factor": 4, "hidden_sizes": [8, 8], "num_layers": 2, "p_dropout": 0.0, "structure": "Stacked"}
    else:
        cfg = {"name": "dcn", "num_factor": 4, "hidden_sizes": [8], "num_layers": 3, "p_dropout": 0.0, "structure": "Parallel"}
    model = pkg.get_ctr_model(dims, dict(cfg))
    missing, unexpected = model.load_state_dict(p, strict=True)   # reference (uncompiled) state_dict keys
    model.to(DEV).train(bool(g["training"]))
    x, y = g.t("x").to(DEV), g.t("y").to(DEV)
    logits = model(x)
    assert_close(logits, g.t("logits"), 1e-4, 1e-5, "logits")
    torch.nn.BCEWithLogitsLoss()(logits, y).backward()
    named = dict(model.named_parameters())
    for k, ref in g.group("grad/").items():
        assert_close(named[k].grad, ref, 5e-4, 2e-5, k)
    pkg.check_index_errors()
    if bool(g["training"]):
        # the same step with the labels handed to the forward and the package's criterion: where the model's tail ends in the
        # fused head (DCN-Mix, stacked DCNv2) head + criterion + head backward are one launch; same logits, same gradients
        from recsys_benchmark_amd.losses import BCEWithLogitsLoss, unit_scalar

        model.zero_grad(set_to_none=True)
        logits2 = model(x, labels=y)
        assert_close(logits2, g.t("logits"), 1e-4, 1e-5, "logits (labels in forward)")
        loss = BCEWithLogitsLoss()(logits2, y)
        want = torch.nn.functional.binary_cross_entropy_with_logits(g.t("logits"), g.t("y"))
        assert abs(float(loss) - float(want)) <= 1e-5
        loss.backward(unit_scalar(DEV))
        for k, ref in g.group("grad/").items():
            assert_close(named[k].grad, ref, 5e-4, 2e-5, k + " (labels in forward)")
        pkg.check_index_errors()


@pytest.mark.parametrize("M,N,E", [(4096, 352, 4), (37, 8, 1), (130, 416, 8), (5, 1024, 3)])
def test_rowdot_multi_is_the_gate_product(M, N, E):
    import ctypes  # noqa: F401

    from recsys_benchmark_amd import _lib

    gen = torch.Generator().manual_seed(M + N + E)
    X, W = torch.randint(-3, 4, (M, N), generator=gen).float(), torch.randint(-3, 4, (E, N), generator=gen).float()
    out = torch.empty(M, E, device=DEV)
    Xd, Wd = X.to(DEV), W.to(DEV)
    _lib.check(_lib.load().mi_rowdot_multi(Xd.data_ptr(), N, Wd.data_ptr(), out.data_ptr(), M, N, E,
                                           _lib.stream_ptr(out.device)), "mi_rowdot_multi")
    assert torch.equal(out.cpu(), X @ W.t())          # integer-valued: exact
    with pytest.raises(_lib.MI355XLibraryError):      # N % 4 != 0 is outside the kernel: the caller takes the GEMM
        _lib.check(_lib.load().mi_rowdot_multi(Xd.data_ptr(), N, Wd.data_ptr(), out.data_ptr(), M, N - 1, E,
                                               _lib.stream_ptr(out.device)), "mi_rowdot_multi")


@pytest.mark.parametrize("M,N", [(4096, 352), (4096, 416), (33, 8), (257, 260), (70, 1024), (1, 4)])
@pytest.mark.parametrize("mix", [True, False])
def test_cross_bwd_head_equals_its_three_passes(M, N, mix):
    """mi_cross_bwd_head (one pass over the rows) against dlin = g*x0, dx0 (+)= g*lin, db = sum_m dlin*rs, dgs = dlin.b
    written out in torch; integer-valued data, so also the float-atomic column sums are exact."""
    from recsys_benchmark_amd import _lib

    gen = torch.Generator().manual_seed(M * 3 + N + mix)
    mk = lambda *s: torch.randint(-3, 4, s, generator=gen).float()          # noqa: E731
    g, x0, lin, b, gate, dx_prev = mk(M, N), mk(M, N), mk(M, N), mk(N), mk(M, 4), mk(M, N)
    lib = _lib.load()
    for accumulate in (0, 1):
        d = lambda t: t.to(DEV)                                             # noqa: E731
        dlin, dx0 = torch.empty(M, N, device=DEV), d(dx_prev).clone()
        db, dgs = torch.zeros(N, device=DEV), torch.empty(M, device=DEV)
        gd, xd, ld, bd, gtd = d(g), d(x0), d(lin), d(b), d(gate)
        _lib.check(lib.mi_cross_bwd_head(gd.data_ptr(), xd.data_ptr(), ld.data_ptr(), gtd.data_ptr() if mix else None, 4,
                                         bd.data_ptr() if mix else None, dlin.data_ptr(), dx0.data_ptr(), accumulate,
                                         db.data_ptr(), dgs.data_ptr() if mix else None, M, N, _lib.stream_ptr(dlin.device)),
                   "mi_cross_bwd_head")
        ref_dlin = g * x0
        rs = gate.sum(1, keepdim=True) if mix else torch.ones(M, 1)
        assert torch.equal(dlin.cpu(), ref_dlin)
        assert torch.equal(dx0.cpu(), (dx_prev if accumulate else 0) + g * lin)
        assert torch.equal(db.cpu(), (ref_dlin * rs).sum(0))
        if mix:
            assert torch.equal(dgs.cpu(), ref_dlin @ b)


@pytest.mark.parametrize("M,d,E,r", [(4096, 352, 4, 64), (200, 40, 3, 16), (67, 8, 1, 32), (130, 416, 2, 64), (1, 4, 4, 16)])
def test_mix_expert_kernels_vs_float64(M, d, E, r):
    """mi_mix_expert_fwd / _bwd (the r x r product in the epilogue of the d-long one) against the same chain written out in
    float64: H1 = tanh(x V_e), H2 = tanh(H1 C_e), H2g = H2 g_e;  dH = dT U_e^T, dgate = sum dH*H2 + dgs,
    dZ2 = dH g_e (1 - H2^2), dZ1 = (dZ2 C_e^T)(1 - H1^2)."""
    from recsys_benchmark_amd import _lib

    gen = torch.Generator().manual_seed(M + d + E + r)
    R = lambda *s: torch.randn(*s, generator=gen)                                  # noqa: E731
    x, dT, gate, dgs = R(M, d) * 0.5, R(M, d), R(M, E), R(M)
    V, C, U = R(E, d, r) / d ** 0.5, R(E, r, r) / r ** 0.5, R(E, r, d) / r ** 0.5
    dev = torch.device(DEV, 0)
    g = lambda t: t.to(DEV)                                                        # noqa: E731
    xd, dTd, gd, dgsd, Vd, Cd, Ud = g(x), g(dT), g(gate), g(dgs), g(V), g(C), g(U)
    H1, H2, H2g = (torch.empty(M, E * r, device=DEV) for _ in range(3))
    lib, st = _lib.load(), _lib.stream_ptr(dev)
    _lib.check(lib.mi_mix_expert_fwd(xd.data_ptr(), Vd.data_ptr(), Cd.data_ptr(), gd.data_ptr(), H1.data_ptr(), H2.data_ptr(),
                                     H2g.data_ptr(), M, d, E, r, st), "fwd")
    x64, V64, C64, U64 = x.double(), V.double(), C.double(), U.double()
    h1 = torch.tanh(torch.einsum("md,edr->mer", x64, V64))
    h2 = torch.tanh(torch.einsum("mek,ekc->mec", h1, C64))
    assert_close(H1.view(M, E, r), h1, 2e-5, 2e-6, "H1")
    assert_close(H2.view(M, E, r), h2, 2e-5, 2e-6, "H2")
    assert_close(H2g.view(M, E, r), h2 * gate.double()[:, :, None], 2e-5, 2e-6, "H2g")
    dgate, dZ2, dZ1 = torch.empty(M, E, device=DEV), torch.empty(M, E * r, device=DEV), torch.empty(M, E * r, device=DEV)
    _lib.check(lib.mi_mix_expert_bwd(dTd.data_ptr(), Ud.data_ptr(), Cd.data_ptr(), gd.data_ptr(), H1.data_ptr(), H2.data_ptr(),
                                     dgsd.data_ptr(), dgate.data_ptr(), dZ2.data_ptr(), dZ1.data_ptr(), M, d, E, r, st), "bwd")
    h1f, h2f = H1.view(M, E, r).double().cpu(), H2.view(M, E, r).double().cpu()      # the saved fp32 activations, as the kernel reads them
    dh = torch.einsum("md,erd->mer", dT.double(), U64)
    assert_close(dgate, (dh * h2f).sum(2) + dgs.double()[:, None], 1e-4, 1e-5 * d ** 0.5, "dgate")
    dz2 = dh * gate.double()[:, :, None] * (1 - h2f * h2f)
    assert_close(dZ2.view(M, E, r), dz2, 1e-4, 1e-5, "dZ2")
    dz1 = torch.einsum("mek,eck->mec", dz2, C64) * (1 - h1f * h1f)
    assert_close(dZ1.view(M, E, r), dz1, 1e-4, 1e-5, "dZ1")
    with pytest.raises(_lib.MI355XLibraryError):          # rank 48 is outside the fused form: the caller keeps the separate products
        _lib.check(lib.mi_mix_expert_fwd(xd.data_ptr(), Vd.data_ptr(), Cd.data_ptr(), gd.data_ptr(), H1.data_ptr(), H2.data_ptr(),
                                         H2g.data_ptr(), M, d, E, 48, st), "fwd")
