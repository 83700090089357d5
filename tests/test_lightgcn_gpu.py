"""GPU parity of the LightGCN propagation (fused CSR SpMM kernels) vs the reference goldens and
the oracle.  fp32; only the per-row summation order differs: rtol 1e-5 / atol 1e-6."""
import pytest
import torch

from conftest import assert_close, golden_names, load_golden
from oracle import reference_ops as ro

from recsys_benchmark_amd import _kernels, _lib
from recsys_benchmark_amd.graph_utils import calculate_sparse_graph_adj_norm
from recsys_benchmark_amd.lightgcn import LightGCN, SingleLightGCN

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=[0, 2], ids=["row-per-wave", "slice-phased"])
def _both_spmm_forms(request, monkeypatch):
    """Every test of this file runs on the row-per-wave SpMM and on the task-balanced, slice-phased one (mi_spmm_sliced;
    forced on whatever the operand's size, with 4 KiB column slices so that the small test graphs have many of them)."""
    monkeypatch.setattr(_kernels, "SLICED_SPMM", request.param)
    monkeypatch.setattr(_kernels.CsrPlan, "SLICE_BYTES", 4096)
    _kernels._plans.clear()


def _same_rows(a, b, what=""):
    """Rows computed by the batch-rows-only last layer (always the row-per-wave kernel) against the same rows of a full
    layer: bit for bit when the full layer runs on the row-per-wave kernel too (same order of additions), within float32
    rounding of the row's terms when it runs on the slice-phased kernel (another fixed order)."""
    if _kernels.SLICED_SPMM == 2:
        assert_close(a, b, 2e-5, 2e-6, what)
    else:
        assert torch.equal(a, b), what


def _sample_graph():
    a = load_golden("cf_sample_adj")
    graph = {}
    for u, i in zip(a["edge_user"].tolist(), a["edge_item"].tolist()):
        graph.setdefault(u, []).append(i)
    return a, graph


def test_adj_builder_matches_reference():
    a, graph = _sample_graph()
    adj = calculate_sparse_graph_adj_norm(graph, int(a["num_item"]), int(a["num_user"]))
    assert torch.equal(adj.crow_indices(), a.t("crow")) and torch.equal(adj.col_indices(), a.t("col"))
    assert_close(adj.values(), a.t("val"), 0, 0)


@pytest.mark.parametrize("name", golden_names("lightgcn_L") + golden_names("single_lightgcn_L"))
def test_lightgcn_matches_reference_golden(name):
    g = load_golden(name)
    a, graph = _sample_graph()
    nu, ni, L = int(a["num_user"]), int(a["num_item"]), int(g["num_layers"])
    cls = SingleLightGCN if name.startswith("single") else LightGCN
    model = cls(nu, ni, num_layers=L, hidden_size=16)
    model.load_state_dict(g.group("param/"), strict=True)
    model.to(DEV)
    adj = calculate_sparse_graph_adj_norm(graph, ni, nu).to(DEV)
    ue, ie = model(adj)
    assert_close(ue, g.t("user_emb"), 1e-5, 1e-6, "user_emb")
    assert_close(ie, g.t("item_emb"), 1e-5, 1e-6, "item_emb")
    users, pos, neg = g.t("users").to(DEV), g.t("pos").to(DEV), g.t("neg").to(DEV)
    loss = ro.bpr_loss(ue[users], ie[pos], ie[neg])   # tiny loss stays in PyTorch
    reg = model.get_reg_loss(users, pos, neg)
    assert_close(loss, g.t("bpr"), 1e-5, 1e-6)
    assert_close(reg, g.t("reg"), 1e-5, 1e-5)
    (loss + 1e-3 * reg).backward()
    named = dict(model.named_parameters())
    for k, ref in g.group("grad/").items():
        assert_close(named[k].grad, ref, 1e-4, 1e-7, k)


def _random_graph(U, I, nnz, seed, power=3.0):
    gen = torch.Generator().manual_seed(seed)
    u = torch.randint(0, U, (nnz,), generator=gen)
    i = (I * torch.rand(nnz, generator=gen).pow(power)).long().clamp_(max=I - 1)   # item power law -> hubs
    n = U + I
    idx = torch.stack([torch.cat([u, i + U]), torch.cat([i + U, u])])
    adj = torch.sparse_coo_tensor(idx, torch.ones(2 * nnz), size=(n, n)).coalesce()
    deg = torch.sparse.sum(adj, dim=1).to_dense().clamp_(min=1).pow(-0.5)
    ii = adj.indices()
    vals = adj.values() * deg[ii[0]] * deg[ii[1]]
    return torch.sparse_coo_tensor(ii, vals, size=(n, n)).coalesce().to_sparse_csr()


@pytest.mark.parametrize("U,I,nnz,D,L", [
    (1500, 500, 20000, 64, 3),    # Yelp-like width, has hub rows (> 256 nnz) -> workgroup-per-row kernel
    (50, 70, 300, 16, 2),
    (40, 30, 200, 7, 2),          # D not a multiple of 4: scalar kernel
    (2000, 3000, 60000, 64, 1),
    (10, 10, 0, 64, 2),           # empty graph: all rows empty
])
@pytest.mark.parametrize("form", ["row-per-wave", "tiled"])
def test_propagate_vs_oracle(U, I, nnz, D, L, form, monkeypatch):
    # both SpMM forms: the row-per-wave kernels (default) and the tiled, LDS-accumulated form (opt-in: correct, slower)
    monkeypatch.setattr(_kernels, "TILED_SPMM", form == "tiled")
    adj = _random_graph(U, I, nnz, seed=U + nnz)
    gen = torch.Generator().manual_seed(1)
    Eu = torch.randn(U, D, generator=gen)
    Ei = torch.randn(I, D, generator=gen)
    G = torch.randn(U + I, D, generator=gen)
    eu, ei = Eu.clone().requires_grad_(True), Ei.clone().requires_grad_(True)
    ref = ro.lightgcn_propagate(adj, torch.cat([eu, ei]), L)
    (ref * G).sum().backward()
    du, di = Eu.to(DEV).requires_grad_(True), Ei.to(DEV).requires_grad_(True)
    out_u, out_i = _kernels.lightgcn_propagate(adj.to(DEV), du, di, L)       # two tables in, two tables out
    out = torch.cat([out_u, out_i])
    assert_close(out, ref, 1e-5, 1e-5, "two-segment forward")
    (out * G.to(DEV)).sum().backward()
    assert_close(du.grad, eu.grad, 1e-4, 1e-5, "grad user table")
    assert_close(di.grad, ei.grad, 1e-4, 1e-5, "grad item table")
    both = torch.cat([Eu, Ei]).to(DEV)
    assert_close(_kernels.lightgcn_propagate(adj.to(DEV), both, None, L), ref, 1e-5, 1e-5, "one-segment forward")
    if nnz:
        plan = _kernels.csr_plan(adj.to(DEV))
        assert plan.pattern_symmetric
        if U == 1500:
            assert plan.long_rows.numel() > 0, "test graph should contain hub rows"


def test_nonsymmetric_values_use_true_transpose():
    # SparseDropout draws independent masks for (i,j) and (j,i): backward must use A^T, not A
    adj = _random_graph(60, 80, 900, seed=4)
    gen = torch.Generator().manual_seed(2)
    vals = adj.values() * (torch.rand(adj.values().shape, generator=gen) < 0.7).float() / 0.7
    adj2 = torch.sparse_csr_tensor(adj.crow_indices(), adj.col_indices(), vals, adj.shape)
    E = torch.randn(140, 16, generator=gen)
    G = torch.randn(140, 16, generator=gen)
    e = E.clone().requires_grad_(True)
    (ro.lightgcn_propagate(adj2, e, 3) * G).sum().backward()
    d = E.to(DEV).requires_grad_(True)
    (_kernels.lightgcn_propagate(adj2.to(DEV), d, None, 3) * G.to(DEV)).sum().backward()
    assert_close(d.grad, e.grad, 1e-4, 1e-5)


def test_rectangular_spmm_and_grad():
    gen = torch.Generator().manual_seed(6)
    idx = torch.stack([torch.randint(0, 30, (200,), generator=gen), torch.randint(0, 45, (200,), generator=gen)])
    A = torch.sparse_coo_tensor(idx, torch.randn(200, generator=gen), size=(30, 45)).coalesce().to_sparse_csr()
    X = torch.randn(45, 64, generator=gen)
    G = torch.randn(30, 64, generator=gen)
    x = X.clone().requires_grad_(True)
    ref = A @ x
    (ref * G).sum().backward()
    d = X.to(DEV).requires_grad_(True)
    out = _kernels.spmm(A.to(DEV), d)
    assert_close(out, ref, 1e-5, 1e-5)
    (out * G.to(DEV)).sum().backward()
    assert_close(d.grad, x.grad, 1e-5, 1e-5)


def test_lightgcn_with_sparse_dropout_and_qr_tables_runs_and_backprops():
    adj = _random_graph(40, 60, 500, seed=8).to(DEV)
    model = LightGCN(40, 60, num_layers=2, hidden_size=16, p_dropout=0.3,
                     embedding_config={"name": "qr", "divider": 4}).to(DEV)
    model.train()
    ue, ie = model(adj)
    assert ue.shape == (40, 16) and ie.shape == (60, 16)
    (ue.sum() + ie.sum()).backward()
    assert model.user_emb_table.emb1.weight.grad is not None
    _lib.check_index_errors()


def test_consecutive_dropout_draws_use_their_own_transposed_values():
    """Every SparseDropout draw is a fresh value tensor that the allocator may place at the address of the previous
    draw: the cached A^T values must never be served for a different draw (gradient vs the oracle on 6 draws)."""
    from recsys_benchmark_amd.layers import SparseDropout

    adj = _random_graph(60, 80, 900, seed=11)
    adj_d = adj.to(DEV)
    drop = SparseDropout(0.5).train()
    gen = torch.Generator().manual_seed(3)
    E = torch.randn(140, 16, generator=gen)
    G = torch.randn(140, 16, generator=gen)
    for it in range(6):
        drawn = drop(adj_d)                      # new values, same pattern -> same CsrPlan
        vals = drawn.values().cpu()
        adj_cpu = torch.sparse_csr_tensor(adj.crow_indices(), adj.col_indices(), vals, adj.shape)
        e = E.clone().requires_grad_(True)
        (ro.lightgcn_propagate(adj_cpu, e, 2) * G).sum().backward()
        d = E.to(DEV).requires_grad_(True)
        (_kernels.lightgcn_propagate(drawn, d, None, 2) * G.to(DEV)).sum().backward()
        assert_close(d.grad, e.grad, 1e-4, 1e-5, f"draw {it}")
        del drawn, d


@pytest.mark.parametrize("U,I,nnz,D,L", [(1500, 500, 20000, 64, 3), (50, 70, 300, 16, 2), (300, 200, 4000, 8, 1),
                                         (90, 110, 900, 4, 2)])      # D = 4: a wave of the mask kernel spans two words
@pytest.mark.parametrize("masked", [True, False])
def test_backward_with_a_gradient_on_few_rows_skips_the_zero_rows_exactly(U, I, nnz, D, L, masked, monkeypatch):
    """The gradient entering a backward propagation is non-zero only on the batch's rows (BPR, L2 reg): the first layer
    of the backward takes a row mask built from that gradient (mi_row_mask) and does not fetch the zero rows
    (mi_spmm_csr_masked).  The mask must equal `any(g != 0, dim=1)` bit for bit, and the gradients must be IDENTICAL to
    the unmasked kernel's (a skipped row contributes +0 either way), for two-segment and one-segment tables, hub rows
    included."""
    monkeypatch.setattr(_kernels, "MASK_FIRST_BACKWARD_LAYER", masked)
    adj = _random_graph(U, I, nnz, seed=3 * U + nnz).to(DEV)
    gen = torch.Generator().manual_seed(5)
    Eu, Ei = torch.randn(U, D, generator=gen), torch.randn(I, D, generator=gen)
    rows = torch.randint(0, U + I, (max(4, (U + I) // 12),), generator=gen)
    G = torch.zeros(U + I, D)
    G[rows] = torch.randn(rows.numel(), D, generator=gen)
    G[rows[0], 1:] = 0.0                                   # a row whose only non-zero is one element
    eu, ei = Eu.clone().requires_grad_(True), Ei.clone().requires_grad_(True)
    (ro.lightgcn_propagate(adj.cpu(), torch.cat([eu, ei]), L) * G).sum().backward()
    du, di = Eu.to(DEV).requires_grad_(True), Ei.to(DEV).requires_grad_(True)
    out_u, out_i = _kernels.lightgcn_propagate(adj, du, di, L)
    (torch.cat([out_u, out_i]) * G.to(DEV)).sum().backward()
    assert_close(du.grad, eu.grad, 1e-4, 1e-5, "grad user table")
    assert_close(di.grad, ei.grad, 1e-4, 1e-5, "grad item table")
    both = torch.cat([Eu, Ei]).to(DEV).requires_grad_(True)
    (_kernels.lightgcn_propagate(adj, both, None, L) * G.to(DEV)).sum().backward()
    assert torch.equal(both.grad, torch.cat([du.grad, di.grad])), "one-segment and two-segment backward agree bit for bit"
    if masked:
        Gd = G.to(DEV)
        want = (Gd != 0).any(dim=1).cpu()
        for a, b in ((Gd[:U].contiguous(), Gd[U:].contiguous()), (Gd, None)):
            words = _kernels._row_mask(a, b, D).cpu()
            bits = ((words.view(-1, 1) >> torch.arange(32, dtype=torch.int32)) & 1).bool().view(-1)[:U + I]
            assert torch.equal(bits, want)
        # and against the unmasked kernel: bit-identical gradients
        monkeypatch.setattr(_kernels, "MASK_FIRST_BACKWARD_LAYER", False)
        ref = torch.cat([Eu, Ei]).to(DEV).requires_grad_(True)
        (_kernels.lightgcn_propagate(adj, ref, None, L) * Gd).sum().backward()
        assert torch.equal(ref.grad, both.grad)


def test_propagation_and_regulariser_as_one_node_equal_the_two_calls():
    """LightGCN.forward_with_reg_loss == (model(adj), model.get_reg_loss(...)): same outputs bit for bit, and the table
    gradients of the full BPR + weight_decay * reg objective within float-atomic ordering (the regulariser's rows are added
    into the propagation's gradient instead of arriving as two dense tensors); repeated ids in the batch included."""
    from recsys_benchmark_amd.losses import bpr_loss_rows

    U, I, D, L, B = 300, 500, 64, 3, 256
    adj = _random_graph(U, I, 6000, seed=9).to(DEV)
    gen = torch.Generator().manual_seed(2)
    users = torch.randint(0, U, (B,), generator=gen).to(DEV)
    pos = torch.randint(0, I, (B,), generator=gen).to(DEV)
    neg = torch.randint(0, 40, (B,), generator=gen).to(DEV)          # many repeats
    grads = []
    for fused in (False, True):
        torch.manual_seed(0)
        model = LightGCN(U, I, num_layers=L, hidden_size=D).to(DEV)
        if fused:
            au, ai, reg = model.forward_with_reg_loss(adj, users, pos, neg)
        else:
            au, ai = model(adj)
            reg = model.get_reg_loss(users, pos, neg)
        loss = bpr_loss_rows(au, ai, users, pos, neg) + 0.05 * reg
        loss.backward()
        grads.append((au.detach(), ai.detach(), reg.detach(), model.user_emb_table.get_weight().grad.clone(),
                      model.item_emb_table.get_weight().grad.clone()))
    for a, b, what in zip(grads[0][:3], grads[1][:3], ("user emb", "item emb", "reg")):
        _same_rows(a, b, what)
    assert_close(grads[1][3], grads[0][3], 1e-5, 1e-7, "user table gradient")
    assert_close(grads[1][4], grads[0][4], 1e-5, 1e-7, "item table gradient")


@pytest.mark.parametrize("L", [1, 3])
def test_last_layer_restricted_to_the_batch_rows_gives_the_same_rows_and_gradients(L):
    """forward_with_reg_loss(..., batch_rows_only=True): the last propagation layer computes only the rows the step reads
    (users / positives / negatives, hubs among them, repeats among them).  Those rows must be bit-identical to the full
    propagation's, and the table gradients of BPR + reg bit-identical too (the backward is the same code on the same
    gradient, which is zero off the batch's rows either way)."""
    from recsys_benchmark_amd.losses import bpr_loss_rows

    U, I, D, B = 1500, 500, 64, 128
    adj = _random_graph(U, I, 20000, seed=21).to(DEV)               # has hub rows (> 256 nnz)
    plan = _kernels.csr_plan(adj)
    assert plan.long_rows.numel() > 0
    hubs_items = plan.long_rows[plan.long_rows >= U].to(torch.int64) - U
    gen = torch.Generator().manual_seed(8)
    users = torch.randint(0, U, (B,), generator=gen).to(DEV)
    pos = torch.randint(0, I, (B,), generator=gen).to(DEV)
    neg = torch.randint(0, 30, (B,), generator=gen).to(DEV)          # many repeats
    if hubs_items.numel():
        pos[:2] = hubs_items[:1]                                        # a hub row in the batch, twice
    outs = []
    for only in (False, True):
        torch.manual_seed(0)
        model = LightGCN(U, I, num_layers=L, hidden_size=D).to(DEV)
        au, ai, reg = model.forward_with_reg_loss(adj, users, pos, neg, batch_rows_only=only)
        (bpr_loss_rows(au, ai, users, pos, neg) + 0.05 * reg).backward()
        outs.append((au.detach()[users], ai.detach()[pos], ai.detach()[neg], reg.detach(),
                     model.user_emb_table.get_weight().grad.clone(), model.item_emb_table.get_weight().grad.clone()))
    for a, b, what in zip(outs[0], outs[1], ("user rows", "positive rows", "negative rows", "reg", "user table grad", "item table grad")):
        if "grad" in what:
            assert_close(b, a, 1e-6, 1e-8, what)      # (BPR / reg rows are scattered with float atomics: order may differ)
        else:
            _same_rows(a, b, what)


def test_fused_step_paths_with_sparse_dropout_reuse_the_plan_across_draws():
    """p_dropout > 0: every forward draws new adjacency values on the same sparsity pattern (one cached plan, new values,
    new transposed values).  The fused call — regulariser in the node, last layer at the batch's rows, masked first
    backward layer, hub flags that must be all zero again after every launch — against the reference's two calls, draw by
    draw under the same seed."""
    from recsys_benchmark_amd.losses import bpr_loss_rows

    U, I, D, B = 1500, 500, 64, 96
    adj = _random_graph(U, I, 20000, seed=31).to(DEV)
    gen = torch.Generator().manual_seed(12)
    users = torch.randint(0, U, (B,), generator=gen).to(DEV)
    pos = torch.randint(0, 20, (B,), generator=gen).to(DEV)            # the hot (hub) items
    neg = torch.randint(0, I, (B,), generator=gen).to(DEV)
    results = []
    for fused in (False, True):
        torch.manual_seed(3)
        model = LightGCN(U, I, num_layers=2, hidden_size=D, p_dropout=0.3).to(DEV).train()
        torch.manual_seed(99)                                           # the dropout draws
        per_draw = []
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            if fused:
                au, ai, reg = model.forward_with_reg_loss(adj, users, pos, neg, batch_rows_only=True)
            else:
                au, ai = model(adj)
                reg = model.get_reg_loss(users, pos, neg)
            (bpr_loss_rows(au, ai, users, pos, neg) + 0.01 * reg).backward()
            per_draw.append((au.detach()[users], ai.detach()[pos], model.user_emb_table.get_weight().grad.clone(),
                             model.item_emb_table.get_weight().grad.clone()))
        results.append(per_draw)
    for d, (a, b) in enumerate(zip(*results)):
        _same_rows(a[0], b[0], f"draw {d}: rows read by the losses")
        _same_rows(a[1], b[1], f"draw {d}: rows read by the losses")
        assert_close(b[2], a[2], 1e-5, 1e-7, f"draw {d}: user table gradient")
        assert_close(b[3], a[3], 1e-5, 1e-7, f"draw {d}: item table gradient")
    for p in _kernels._plans.values():
        assert not bool(p.hub_need.any()), "hub flags are cleared by the launch that consumed them"


@pytest.mark.parametrize("rows_only", [False, True])
def test_single_table_model_fused_step_equals_the_two_calls(rows_only):
    """SingleLightGCN.forward_with_reg_loss (one table, items after the users) == (model(adj), model.get_reg_loss(...)):
    the rows the losses read bit for bit, the table gradient within float-atomic ordering."""
    from recsys_benchmark_amd.losses import bpr_loss_rows

    U, I, D, L, B = 700, 300, 32, 2, 64
    adj = _random_graph(U, I, 9000, seed=41).to(DEV)
    gen = torch.Generator().manual_seed(6)
    users = torch.randint(0, U, (B,), generator=gen).to(DEV)
    pos = torch.randint(0, I, (B,), generator=gen).to(DEV)
    neg = torch.randint(0, 25, (B,), generator=gen).to(DEV)
    outs = []
    for fused in (False, True):
        torch.manual_seed(0)
        model = SingleLightGCN(U, I, num_layers=L, hidden_size=D).to(DEV)
        if fused:
            au, ai, reg = model.forward_with_reg_loss(adj, users, pos, neg, batch_rows_only=rows_only)
        else:
            au, ai = model(adj)
            reg = model.get_reg_loss(users, pos, neg)
        (bpr_loss_rows(au, ai, users, pos, neg) + 0.05 * reg).backward()
        outs.append((au.detach()[users], ai.detach()[pos], ai.detach()[neg], reg.detach(), model.emb_table.get_weight().grad.clone()))
    for a, b, what in zip(outs[0][:3], outs[1][:3], ("user rows", "positive rows", "negative rows")):
        _same_rows(a, b, what)
    assert_close(outs[1][3], outs[0][3], 1e-5, 1e-7, "reg loss (norm^2 vs sum of squares)")
    assert_close(outs[1][4], outs[0][4], 1e-5, 1e-7, "table gradient")
