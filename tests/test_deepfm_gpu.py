"""GPU parity: product DeepFM (HIP gather+FM kernels through the C-ABI) vs the golden
vectors from the reference and vs the oracle on seeded random inputs.

Tolerances (fp32; the only differences are summation order inside the FM reduction,
the float-atomic scatter order of the dense backward and rocBLAS vs CPU GEMM order):
logits rtol 2e-5 / atol 2e-6; gradients rtol 1e-4 / atol 5e-6.  Indices are bit-exact.
"""
import pytest
import torch

from conftest import assert_close, assert_within_terms, golden_names, load_golden
from oracle import reference_ops as ro

import recsys_benchmark_amd as pkg
from recsys_benchmark_amd import _kernels, _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[True, False], ids=["own-tail", "library-tail"])
def _both_tails(request, monkeypatch):
    """Every test of this file runs with the MLP tail on the own fused kernels (the default) and on the general path
    (library products + the fused BatchNorm passes), so both stay covered whatever the default is."""
    from recsys_benchmark_amd import mlp as _mlp_mod

    monkeypatch.setattr(_mlp_mod, "FUSED_TAIL", request.param)
DEV = "cuda"


def _build_from_golden(g, sparse=False, fc_sparse=False):
    p = g.group("param/")
    dims = g["field_dims"].tolist()
    D = p["embedding._emb_module.weight"].shape[1]
    hidden = [p[k].shape[0] for k in sorted(p, key=lambda s: int(s.split(".")[1]) if s.startswith("_deep_branch") else -1)
              if k.startswith("_deep_branch") and k.endswith(".weight") and p[k].dim() == 2][:-1]
    cfg = {"name": "vanilla"}
    if sparse:
        cfg["sparse"] = True
    m = pkg.DeepFM(dims, D, hidden, p_dropout=0.0, use_batchnorm=bool(g["use_bn"]),
                   embedding_config=cfg, fc_sparse=fc_sparse)
    missing, unexpected = m.load_state_dict(p, strict=True)  # reference state_dict keys load as-is
    assert not missing and not unexpected
    return m.to(DEV)


def _can_pack(m):
    D = m.embedding.get_weight().shape[1]
    return D in (4, 8, 16)


@pytest.mark.parametrize("name", golden_names("deepfm_"))
@pytest.mark.parametrize("grad_form", ["dense", "rows"])
@pytest.mark.parametrize("layout", ["split", "packed128"])
def test_deepfm_matches_reference_golden(name, grad_form, layout):
    g = load_golden(name)
    rows = grad_form == "rows"
    m = _build_from_golden(g, sparse=rows, fc_sparse=rows)
    if layout == "packed128":
        # the reference's two tensors as views of ONE [N, 32] buffer (DeepFM.pack_tables): same goldens must hold
        if not _can_pack(m):
            pytest.skip("pack_tables() covers D in {4, 8, 16}")
        m.pack_tables()
        assert m.tables_packed
        sd = m.state_dict()      # ... and the checkpoint keeps the reference's format
        for k in ("embedding._emb_module.weight", "fc.weight"):
            assert sd[k].is_contiguous() and torch.equal(sd[k].cpu(), g.group("param/")[k]), k
    m.train(bool(g["training"]))
    x, y = g.t("x").to(DEV), g.t("y").to(DEV)
    logits = m(x)
    assert_close(logits, g.t("logits"), 2e-5, 2e-6, "logits")
    loss = torch.nn.BCEWithLogitsLoss()(logits, y)
    loss.backward()
    _lib.check_index_errors()
    named = dict(m.named_parameters())
    for k, ref in g.group("grad/").items():
        if k.startswith("linear_layer"):
            continue
        got = named[k].grad
        assert got is not None, k
        if rows and k in ("embedding._emb_module.weight", "fc.weight"):
            assert got.is_sparse
        assert_close(got, ref, 1e-4, 5e-6, f"grad {k}")


def _random_case(B, dims, D, seed, zipf=False):
    gen = torch.Generator().manual_seed(seed)
    N = sum(dims)
    p = {
        "offsets": ro.field_offsets(dims),
        "embedding._emb_module.weight": (torch.rand(N, D, generator=gen) - 0.5),
        "fc.weight": torch.randn(N, 1, generator=gen),
        "_bias": torch.randn(1, generator=gen),
    }
    cols = []
    for d in dims:
        if zipf:
            u = torch.rand(B, generator=gen)
            cols.append((d * u.pow(3)).long().clamp_(max=d - 1))
        else:
            cols.append(torch.randint(0, d, (B,), generator=gen))
    x = torch.stack(cols, 1) if B > 0 else torch.zeros((0, len(dims)), dtype=torch.long)
    g_emb = torch.randn(B, len(dims), D, generator=gen)
    g_y = torch.randn(B, generator=gen)
    return p, x, g_emb, g_y


CASES = [
    # (B, dims, D)                      what it exercises
    (1, [3], 4),                        # single sample, single field, LPR=1
    (7, [5, 7, 11], 4),                 # tiny
    (64, [50] * 26, 16),                # headline shape: LPR=4, NIT=2
    (33, [17] * 39, 16),                # F=39: NIT=3
    (19, [9] * 70, 16),                 # F > 4*RS: generic NIT=0 loop
    (40, [31, 2, 900, 5], 64),          # LightGCN-width rows, LPR=16
    (12, [13, 6], 256),                 # LPR=64, one row per wave-instruction
    (21, [4, 9, 2], 7),                 # D not a multiple of 4: scalar kernels
    (9, [6, 5], 12),                    # D multiple of 4 but LPR=3 not a power of two: scalar kernels
    (5000, [1000, 3, 70000, 12], 8),    # more samples than resident waves: grid-stride
]


@pytest.mark.parametrize("B,dims,D", CASES)
@pytest.mark.parametrize("zipf", [False, True])
@pytest.mark.parametrize("layout", ["split", "packed128", "strided"])
def test_gather_fm_kernels_vs_oracle(B, dims, D, zipf, layout):
    p, x, g_emb, g_y = _random_case(B, dims, D, seed=B * 131 + D, zipf=zipf)
    for v in p.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    emb_ref, y_ref = ro.deepfm_embed_fm(x, p)
    ((emb_ref * g_emb).sum() + (y_ref.squeeze(1) * g_y).sum()).backward()

    W = p["embedding._emb_module.weight"].detach().to(DEV)
    w1 = p["fc.weight"].detach().to(DEV)
    if layout != "split":
        # both tables as column slices of one buffer: [N, 32] = {row, w1, pad} (pack_tables' layout), or an odd layout
        # (row stride D + 8, w1 in a buffer of its own with stride 3) that only the stride arguments describe
        if D % 4 or (layout == "packed128" and D > 16):
            pytest.skip("row-strided tables need float4 rows" if D % 4 else "packed128 holds D <= 16")
        N = W.shape[0]
        if layout == "packed128":
            buf = torch.zeros(N, 32, device=DEV)
            buf[:, :D], buf[:, D:D + 1] = W, w1
            W, w1 = buf[:, :D], buf[:, D:D + 1]
        else:
            bw, b1 = torch.full((N, D + 8), 7.0, device=DEV), torch.full((N, 3), 7.0, device=DEV)
            bw[:, :D], b1[:, 1:2] = W, w1
            W, w1 = bw[:, :D], b1[:, 1:2]
        assert not W.is_contiguous() or N <= 1
    W = W.detach().requires_grad_(True)
    w1 = w1.detach().requires_grad_(True)
    bias = p["_bias"].detach().to(DEV).requires_grad_(True)
    off = p["offsets"].to(DEV)
    for sparse in (False, True):
        W.grad = w1.grad = bias.grad = None
        emb, yfm = _kernels.gather_fm(x.to(DEV), off, W, w1, bias, sparse_W=sparse, sparse_w1=sparse)
        assert torch.equal(emb.cpu(), emb_ref.detach()), "gathered rows must be exact copies"
        assert_close(yfm, y_ref.squeeze(1), 2e-5, 2e-5, "y_fm")
        ((emb * g_emb.to(DEV)).sum() + (yfm * g_y.to(DEV)).sum()).backward()
        # a hot row sums up to B addends of magnitude ~10 in an order that differs from the
        # CPU's (float atomics / coalesce): the absolute tolerance scales with the addend count
        atol = 1e-5 + 2e-7 * B * 10
        assert_close(W.grad, p["embedding._emb_module.weight"].grad, 1e-4, atol, f"gW sparse={sparse}")
        assert_close(w1.grad, p["fc.weight"].grad, 1e-4, atol, f"gw1 sparse={sparse}")
        assert_close(bias.grad, p["_bias"].grad, 1e-4, atol, "gbias")
    _lib.check_index_errors()


@pytest.mark.parametrize("B,dims,D", [c for c in CASES if c[0] >= 2 and c[2] % 4 == 0 and (c[2] // 4) & (c[2] // 4 - 1) == 0])
@pytest.mark.parametrize("layout", ["split", "packed128"])
@pytest.mark.parametrize("fc_sparse", [True, False])
def test_lookup_backward_in_the_dgrad_epilogue_vs_oracle_and_vs_the_two_node_path(B, dims, D, layout, fc_sparse, monkeypatch):
    """The gather + FM backward folded into the epilogue of the tail's first input-gradient product (tail.DeepFMFusedFn,
    mi_tail_dgrad_gemm_fm): logits and EVERY gradient of a whole DeepFM step against the oracle
    (oracle/reference_ops.deepfm_forward = src/models/deepfm.py:79-105 through autograd) at the gather kernels' CASES,
    and against this build's two-node path (mi_gather_fm_bwd_rows behind the tail) on the same inputs.  fc_sparse False =
    the reference's sparse config (configs/deepfm/base_config_sparse.yaml: row-form embedding gradient, dense first-order)."""
    from recsys_benchmark_amd import mlp as _mlp_mod, tail as _tail_mod

    if layout == "packed128" and D > 16:
        pytest.skip("packed128 holds D <= 16")
    F = len(dims)
    torch.manual_seed(B + D)
    hidden = [32, 16]
    base = pkg.DeepFM(dims, D, hidden, p_dropout=0.0, use_batchnorm=True, embedding_config={"name": "vanilla", "sparse": True},
                      fc_sparse=fc_sparse)
    with torch.no_grad():
        base._bias.fill_(0.3)
        for m in base._deep_branch:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    gen = torch.Generator().manual_seed(B * 7 + D)
    x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1)
    y = (torch.rand(B, generator=gen) < 0.3).float()
    p = {k: v.detach().clone() for k, v in base.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    fits = _mlp_mod.FUSED_TAIL and (F * D) % 8 == 0 and B >= 2
    if B >= 2:
        ref = ro.deepfm_forward(x, p, len(hidden), True, True)
        torch.nn.BCEWithLogitsLoss()(ref, y).backward()

    import copy
    got = {}
    for fused in (True, False):
        monkeypatch.setattr(_tail_mod, "FM_EPILOGUE", fused)
        m = copy.deepcopy(base).to(DEV).train()
        if layout == "packed128":
            m.pack_tables()
        logits = m(x.to(DEV))
        assert (type(logits.grad_fn.next_functions[0][0]).__name__ == "DeepFMFusedFnBackward") == (fused and fits)
        torch.nn.BCEWithLogitsLoss()(logits, y.to(DEV)).backward()
        _lib.check_index_errors()
        grads = {k: (v.grad.to_dense() if v.grad.is_sparse else v.grad).cpu() for k, v in m.named_parameters() if v.grad is not None}
        assert m.embedding.get_weight().grad.is_sparse and m.fc.weight.grad.is_sparse == fc_sparse
        got[fused] = (logits.detach().cpu(), grads)
    if B < 2:
        return          # (BatchNorm needs two samples; the reference raises there)
    atol = 1e-5 + 2e-7 * B
    for fused in (True, False):
        logits, grads = got[fused]
        assert_close(logits, ref.detach(), 1e-4, 1e-5, f"logits fused={fused}")
        for k, gr in grads.items():
            if k.startswith("linear_layer") or (k.endswith(".bias") and k.startswith("_deep_branch") and p[k].grad.abs().max() < 1e-6):
                continue
            assert_close(gr, p[k].grad, 2e-4, atol, f"grad {k} fused={fused}")
    # the two paths share every forward kernel: same logits up to the order of the float atomics that accumulate the
    # BatchNorm statistics above one 64-row tile; table gradients equal up to the order in which a row's duplicates are
    # summed when densified.  (B = 5000 is 79 tiles met by atomics in an order that changes from launch to launch: a logit
    # of 0.055 was seen 1.7e-6 apart between the two paths — both well inside the 1e-5 against the oracle above)
    assert_close(got[True][0], got[False][0], 1e-5, 5e-6, "logits: fused epilogue vs two-node path")
    for k in ("embedding._emb_module.weight", "fc.weight", "_bias"):
        assert_close(got[True][1][k], got[False][1][k], 1e-5, atol, f"{k}: fused epilogue vs two-node path")


@pytest.mark.parametrize("B,hidden,p_drop", [(64, [32, 16], 0.0), (257, [64, 32], 0.5), (4096, [400, 400, 400], 0.5), (1000, [512], 0.0)])
@pytest.mark.parametrize("layout", ["split", "packed128"])
def test_criterion_in_the_head_launch_vs_the_three_launch_form_and_the_oracle(B, hidden, p_drop, layout, monkeypatch):
    """DeepFM.forward(x, labels=y) + BCEWithLogitsLoss(logits, y): head, criterion and the head's backward sums in ONE launch
    (mi_tail_head_bce), the masks and the zero fill carried by the gather launch (mi_gather_fm_fwd_ride) — against the same
    step with the labels withheld (three launches: mi_tail_head_fwd_m, mi_bce_logits_fwd, mi_tail_head_bwd_s) on the same
    dropout seed, and (without dropout) against the oracle through autograd; an upstream gradient other than the resident 1
    must take the general path and still be right."""
    import copy
    from recsys_benchmark_amd import mlp as _mlp_mod, tail as _tail_mod
    from recsys_benchmark_amd.losses import BCEWithLogitsLoss, unit_scalar

    dims, D = [7, 3, 50, 11, 4, 200, 9, 31], 16
    torch.manual_seed(B + len(hidden))
    base = pkg.DeepFM(dims, D, hidden, p_dropout=p_drop, use_batchnorm=True, embedding_config={"name": "vanilla", "sparse": True},
                      fc_sparse=True)
    with torch.no_grad():
        base._bias.fill_(-0.2)
        for m in base._deep_branch:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    gen = torch.Generator().manual_seed(B * 3 + 1)
    x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1)
    y = (torch.rand(B, generator=gen) < 0.3).float()
    dev = torch.device(DEV, 0)
    lossf = BCEWithLogitsLoss()

    x_next = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1).to(DEV)

    def run(labels_in_forward, upstream):
        m = copy.deepcopy(base).to(DEV).train()
        if layout == "packed128":
            m.pack_tables()
        _mlp_mod._seed_word(dev).fill_(4242)
        xd, yd = x.to(DEV), y.to(DEV)
        if labels_in_forward:      # the step also knows the NEXT batch: its weight-gradient launch carries the prefetch riders
            m.prefetch_next(x_next)
        logits = m(xd, labels=yd) if labels_in_forward else m(xd)
        loss = lossf(logits, yd)
        fused = type(loss.grad_fn).__name__ == "_HeadBCEFnBackward"
        assert fused == (labels_in_forward and _mlp_mod.FUSED_TAIL)      # (the library-tail runs have no head launch to ride in)
        if upstream is None:
            loss.backward(unit_scalar(dev))
        else:
            (loss * upstream).backward()
        _lib.check_index_errors()
        grads = {k: (v.grad.to_dense() if v.grad.is_sparse else v.grad).cpu() for k, v in m.named_parameters() if v.grad is not None}
        stats = {k: v.detach().cpu() for k, v in m.state_dict().items() if "running_" in k}
        return logits.detach().cpu(), float(loss), grads, stats

    one = run(True, None)
    three = run(False, None)
    atol = 1e-5 + 2e-7 * B
    # (two runs of one step differ by the order of the float atomics behind the BatchNorm statistics: a few 1e-6 on O(1) logits)
    assert_close(one[0], three[0], 1e-5, 1e-5, "logits")
    assert abs(one[1] - three[1]) <= 1e-6 + 1e-5 * abs(three[1]), (one[1], three[1])
    assert set(one[2]) == set(three[2])
    for k in three[2]:
        assert_close(one[2][k], three[2][k], 2e-4, atol, f"grad {k}: one launch vs three")
    for k in three[3]:
        assert_close(one[3][k], three[3][k], 1e-5, 1e-6, f"{k}")
    scaled = run(True, 2.5)       # not the resident 1: general backward of the criterion, the model's own head backward
    for k in three[2]:
        assert_close(scaled[2][k], 2.5 * three[2][k], 2e-4, 2.5 * atol, f"grad {k}: upstream 2.5")
    if p_drop == 0.0:
        p = {k: v.detach().clone() for k, v in base.state_dict().items()}
        for k, v in p.items():
            if v.is_floating_point() and "running_" not in k:
                v.requires_grad_(True)
        ref = ro.deepfm_forward(x, p, len(hidden), True, True)
        ref_loss = torch.nn.BCEWithLogitsLoss()(ref, y)
        ref_loss.backward()
        assert_close(one[0], ref.detach(), 1e-4, 1e-5, "logits vs oracle")
        assert abs(one[1] - float(ref_loss)) <= 1e-5, (one[1], float(ref_loss))
        for k, gr in one[2].items():
            if k.startswith("linear_layer") or (k.endswith(".bias") and k.startswith("_deep_branch") and p[k].grad.abs().max() < 1e-6):
                continue
            assert_close(gr, p[k].grad, 2e-4, atol, f"grad {k} vs oracle")
    # labels that are not the ones the forward saw: the criterion must not pick the head launch's result up
    m = copy.deepcopy(base).to(DEV).train()
    logits = m(x.to(DEV), labels=y.to(DEV))
    other = (1.0 - y).to(DEV)
    loss = lossf(logits, other)
    assert type(loss.grad_fn).__name__ == "_BCEFnBackward"
    want = torch.nn.functional.binary_cross_entropy_with_logits(logits.detach(), other)
    assert abs(float(loss) - float(want)) <= 1e-5
    loss.backward()
    assert all(torch.isfinite(v.grad.to_dense() if v.grad.is_sparse else v.grad).all() for v in m.parameters() if v.grad is not None)


@pytest.mark.parametrize("B,hidden,bn", [(64, [400, 400, 400], True), (4096, [400, 400, 400], True), (300, [64, 32], False), (7, [32], True)])
@pytest.mark.parametrize("layout", ["split", "packed128"])
def test_inference_forward_is_the_lookup_and_the_products_only(B, hidden, bn, layout):
    """model.eval() under no_grad (the reference's scripts/deepfm/infer_deepfm.py:318-352, validate_epoch): DeepFM's forward
    as the lookup launch — carrying the constants of the tail's fixed-statistics layers and the logits' zero fill in extra
    workgroups — and one launch per hidden layer, the head Linear(., 1) in the epilogue of the last one.  Against the oracle's
    eval forward; the launches are read off the dispatch-event ring."""
    from recsys_benchmark_amd import mlp as _mlp_mod
    from recsys_benchmark_amd.profiling import KernelTimer

    dims, D = [7, 3, 50, 11, 4, 200, 9, 31], 16
    torch.manual_seed(B + len(hidden))
    model = pkg.DeepFM(dims, D, hidden, p_dropout=0.5, use_batchnorm=bn)
    with torch.no_grad():
        model._bias.fill_(0.15)
        for m in model._deep_branch:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(B)
    x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1)
    ref = ro.deepfm_forward(x, p, len(hidden), bn, False)
    model = model.to(DEV).eval()
    if layout == "packed128":
        model.pack_tables()
    with torch.no_grad():
        model(x.to(DEV))                                  # (first call: lazily created buffers)
        with KernelTimer(32) as kt:
            logits = model(x.to(DEV))
    names = [k for k, _ in kt.records]
    if _mlp_mod.FUSED_TAIL:
        assert names[0] == "gather_fm_fwd_ride" and len(names) == 1 + len(hidden), names
        assert all(n in ("tail_fwd_gemm", "tail_fwd_gemm_small") for n in names[1:]), names
    assert_close(logits, ref.detach(), 1e-4, 1e-5, "eval logits")
    _lib.check_index_errors()


def test_empty_batch():
    p, x, g_emb, g_y = _random_case(0, [5, 6], 16, seed=1)
    emb, yfm = _kernels.gather_fm(x.to(DEV), p["offsets"].to(DEV), p["embedding._emb_module.weight"].to(DEV),
                                  p["fc.weight"].to(DEV), p["_bias"].to(DEV))
    assert emb.shape == (0, 2, 16) and yfm.shape == (0,)


def test_out_of_range_index_is_flagged_not_faulting():
    p, x, _, _ = _random_case(8, [5, 6], 16, seed=2)
    x[3, 1] = 6 + 1000           # beyond the last row of the table
    x[5, 0] = -12                # negative
    emb, yfm = _kernels.gather_fm(x.to(DEV), p["offsets"].to(DEV), p["embedding._emb_module.weight"].to(DEV),
                                  p["fc.weight"].to(DEV), p["_bias"].to(DEV))
    torch.cuda.synchronize()
    assert torch.count_nonzero(emb[3, 1]) == 0 and torch.count_nonzero(emb[5, 0]) == 0
    with pytest.raises(IndexError):
        _lib.check_index_errors()
    _lib.check_index_errors()    # flag was cleared


def test_int32_indices_and_noncontiguous_inputs():
    p, x, _, _ = _random_case(16, [5, 6, 7], 16, seed=3)
    ref_emb, ref_y = ro.deepfm_embed_fm(x, p)
    xt = x.to(torch.int32).to(DEV).t().contiguous().t()   # non-contiguous int32 view
    emb, yfm = _kernels.gather_fm(xt, p["offsets"].to(DEV), p["embedding._emb_module.weight"].to(DEV),
                                  p["fc.weight"].to(DEV), p["_bias"].to(DEV))
    assert torch.equal(emb.cpu(), ref_emb)
    assert_close(yfm, ref_y.squeeze(1), 2e-5, 2e-5)


@pytest.mark.parametrize("n,N,D", [(1, 3, 4), (100, 50, 16), (1000, 7, 64), (37, 11, 7), (5000, 100000, 16), (0, 5, 16)])
@pytest.mark.parametrize("shape2d", [False, True])
def test_vanilla_embedding_lookup_and_grad(n, N, D, shape2d):
    gen = torch.Generator().manual_seed(n + N)
    emb = pkg.VanillaEmbedding(N, D).to(DEV)
    W = emb.get_weight()
    idx = torch.randint(0, N, (n,), generator=gen)
    if shape2d and n % 2 == 0 and n > 0:
        idx = idx.view(n // 2, 2)
    out = emb(idx.to(DEV))
    ref = torch.nn.functional.embedding(idx, W.detach().cpu())
    assert torch.equal(out.cpu(), ref)
    G = torch.randn(out.shape, generator=gen)
    (out * G.to(DEV)).sum().backward()
    Wc = W.detach().cpu().requires_grad_(True)
    (torch.nn.functional.embedding(idx, Wc) * G).sum().backward()
    # hot rows sum ~150 unit-size terms in a run-dependent order (float atomics): the error follows sum|terms|, not the
    # result — asserted as such against the float64 sum, with the stock CPU float32 result held to the same bound
    flat = idx.reshape(-1)
    ref64 = torch.zeros(N, D, dtype=torch.float64).index_add_(0, flat, G.reshape(-1, D).double())
    terms = torch.zeros(N, D, dtype=torch.float64).index_add_(0, flat, G.reshape(-1, D).double().abs())
    assert_within_terms(W.grad, ref64, terms, 8, "dense scatter-add", cpu32=Wc.grad)
    assert_close(W.grad, Wc.grad, 1e-5, 1e-4, "dense scatter-add")
    semb = pkg.VanillaEmbedding(N, D, sparse=True).to(DEV)
    out = semb(idx.to(DEV))
    (out * G.to(DEV)).sum().backward()
    assert semb.get_weight().grad.is_sparse
    Wc2 = semb.get_weight().detach().cpu().requires_grad_(True)
    (torch.nn.functional.embedding(idx, Wc2) * G).sum().backward()
    assert_within_terms(semb.get_weight().grad, ref64, terms, 8, "row-form grad (coalesced)", cpu32=Wc2.grad)
    assert_close(semb.get_weight().grad, Wc2.grad, 1e-5, 1e-4, "row-form grad")


@pytest.mark.parametrize("mode", ["sum", "mean", "max"])
def test_vanilla_embedding_bag_modes(mode):
    gen = torch.Generator().manual_seed(5)
    emb = pkg.VanillaEmbedding([5, 6], 16, mode=mode).to(DEV)
    idx = torch.randint(0, 11, (9, 4), generator=gen)
    ref = torch.nn.functional.embedding_bag(idx, emb.get_weight().detach().cpu(), mode=mode)
    assert_close(emb(idx.to(DEV)), ref, 1e-6, 1e-6)


def test_deepfm_over_a_qat_table_uses_the_rounded_rows():
    """`embedding_config: {name: qat}`: the fused vanilla gather+FM kernel must NOT be taken for the QAT subclass —
    the FM and the MLP see stochastic-rounded rows (src/models/embeddings/qat_emb.py:117-119)."""
    import recsys_benchmark_amd as pkg

    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    m = pkg.DeepFM([11, 7, 5], 8, [16], p_dropout=0.0, use_batchnorm=False,
                   embedding_config={"name": "qat", "n_bits": 8}).to(dev)
    x = torch.stack([torch.randint(0, d, (32,)) for d in (11, 7, 5)], 1).to(dev)
    seen = {}
    def remember(mod, inp, out):          # (a hook that returns a value would replace the module's output)
        seen["emb"] = out.detach()

    hook = m.embedding.register_forward_hook(remember)
    m(x).sum().backward()
    hook.remove()
    s = float(m.embedding.scale.detach())
    q = seen["emb"] / s
    assert float((q - torch.round(q)).abs().max()) < 1e-3, "rows on the quantisation grid"
    assert m.embedding.scale.grad is not None and m.embedding._emb_module.weight.grad is not None


def test_deterministic_mode_gives_bit_identical_steps_and_the_reference_gradients():
    """use_deterministic_algorithms(True): the reference-default DENSE weight.grad (a deterministic index_add on the CPU,
    src/models/embeddings/base.py:74-75) built by sorted accumulation instead of float atomics, the MLP tail on the
    atomic-free fused kernels: two runs of the same step agree BITWISE on every gradient, hot rows included, and the
    gradients are the oracle's."""
    import copy

    dims = [3, 40, 7, 1000, 2]           # fields of 2 and 3 values: ~B/2 duplicates per row
    torch.manual_seed(5)
    base = pkg.DeepFM(dims, 16, [64, 32], p_dropout=0.0, use_batchnorm=True).to(DEV)
    gen = torch.Generator().manual_seed(9)
    x = torch.stack([torch.randint(0, d, (2048,), generator=gen) for d in dims], 1).to(DEV)
    y = (torch.rand(2048, generator=gen) < 0.3).float().to(DEV)
    pkg.use_deterministic_algorithms(True)
    try:
        runs = []
        for _ in range(2):
            m = copy.deepcopy(base).train()
            torch.nn.functional.binary_cross_entropy_with_logits(m(x), y).backward()
            runs.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        assert runs[0].keys() == runs[1].keys() and "embedding._emb_module.weight" in runs[0]
        for k in runs[0]:
            assert not runs[0][k].is_sparse
            assert torch.equal(runs[0][k], runs[1][k]), f"{k} differs between two runs of the same step"
    finally:
        pkg.use_deterministic_algorithms(False)
    p = {k: v.detach().cpu().clone() for k, v in base.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    torch.nn.functional.binary_cross_entropy_with_logits(ro.deepfm_forward(x.cpu(), p, 2, True, True), y.cpu()).backward()
    assert_close(runs[0]["embedding._emb_module.weight"], p["embedding._emb_module.weight"].grad, 1e-4, 1e-7, "table grad")
    assert_close(runs[0]["fc.weight"], p["fc.weight"].grad, 1e-4, 1e-7, "first-order grad")
    _lib.check_index_errors()
