"""GPU parity of the compressed-embedding plug-ins (QR, CERP, CSR-pruned, DHE) against the
reference's golden vectors and the oracle.  Index math is bit-exact; rows that are copies or
single fp32 ops of table entries must match exactly; gradients (float-atomic sums) to
rtol 1e-5 / atol 1e-6."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_close, golden_names, load_golden
from oracle import int_ops
from oracle import reference_ops as ro

import recsys_benchmark_amd as pkg
from recsys_benchmark_amd import _kernels, _lib
from recsys_benchmark_amd.embeddings import (CerpEmbedding, DHEmbedding, PrunedEmbedding, QRHashingEmbedding,
                                             RetrainCerpEmbedding, get_embedding)

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ------------------------------------------------------------------ QR
@pytest.mark.parametrize("name", golden_names("qr_"))
def test_qr_matches_reference_golden(name):
    g = load_golden(name)
    op = name.split("_")[1]
    cfg = {"name": "qr", "operation": op}
    if not name.endswith("None"):
        cfg["divider"] = int(g["divider"])
    emb = get_embedding(cfg, g["field_dims"].tolist(), int(g["hidden"]))
    assert emb._divider == int(g["divider"])
    emb.load_state_dict(g.group("param/"), strict=True)
    emb.to(DEV)
    for tag in ("1d", "2d"):
        emb.zero_grad()
        out = emb(g.t(f"x_{tag}").to(DEV))
        assert_close(out, g.t(f"out_{tag}"), 0, 0, f"out {tag}")
        (out * g.t(f"G_{tag}").to(DEV)).sum().backward()
        assert_close(emb.emb1.weight.grad, g.t(f"grad_{tag}/emb1.weight"), 1e-5, 1e-6, "g emb1")
        assert_close(emb.emb2.weight.grad, g.t(f"grad_{tag}/emb2.weight"), 1e-5, 1e-6, "g emb2")
    assert_close(emb.get_weight(), g.t("weight"), 0, 0, "get_weight")
    _lib.check_index_errors()


@pytest.mark.parametrize("op", ["mult", "add", "cat"])
@pytest.mark.parametrize("N,D,divider,shape", [
    (1000, 16, 2, (64, 22)),       # Avazu-like: divider 2 -> a 2-row remainder table (LDS pre-sum path)
    (100003, 16, None, (128, 5)),  # sqrt divider, big quotient table (direct atomics)
    (50, 6, 7, (9,)),              # row width not a multiple of 4 / cat halves of 3: scalar kernels
    (4096, 64, 64, (300,)),        # LightGCN-width rows
])
def test_qr_vs_oracle(op, N, D, divider, shape):
    if op == "cat" and D % 2:
        pytest.skip("cat needs an even width")
    gen = torch.Generator().manual_seed(N + D)
    emb = QRHashingEmbedding(N, D, None, divider, op).to(DEV)
    idx = torch.randint(0, N, shape, generator=gen)
    out = emb(idx.to(DEV))
    e1 = emb.emb1.weight.detach().cpu().requires_grad_(True)
    e2 = emb.emb2.weight.detach().cpu().requires_grad_(True)
    ref = ro.qr_forward(idx, e1, e2, emb._divider, op)
    assert_close(out, ref, 0, 0, "forward")
    G = torch.randn(ref.shape, generator=gen)
    (ref * G).sum().backward()
    (out * G.to(DEV)).sum().backward()
    assert_close(emb.emb1.weight.grad, e1.grad, 2e-5, 2e-5, "g emb1")
    assert_close(emb.emb2.weight.grad, e2.grad, 2e-5, 2e-5, "g emb2")


@pytest.mark.parametrize("op", ["mult", "add", "cat"])
def test_qr_row_form_gradient_of_the_quotient_table_is_the_dense_one(op):
    """QRHashingEmbedding(sparse=True) (an extension, like sparse=True on the DeepFM tables): the quotient table's gradient
    as uncoalesced COO rows (mi_dual_gather_bwd_rows) — densified it is the dense path's gradient; EXACT on integer-valued
    data (no rounding, so the order of additions cannot matter), the remainder table's dense gradient likewise."""
    gen = torch.Generator().manual_seed(11)
    dims = [241, 8, 3697, 5, 31]
    N, D, B = sum(dims), 16, 512
    off = torch.tensor([0] + dims[:-1]).cumsum(0)
    idx = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1) + off
    res = {}
    for sparse in (False, True):
        torch.manual_seed(3)
        gen = torch.Generator().manual_seed(12)          # the same integer tables for both forms
        emb = QRHashingEmbedding(dims, D, None, 2, op, sparse=sparse).to(DEV)
        with torch.no_grad():
            for w in (emb.emb1.weight, emb.emb2.weight):
                w.copy_(torch.randint(-3, 4, w.shape, generator=gen).float())
        out = emb(idx.to(DEV))
        G = torch.randint(-2, 3, out.shape, generator=torch.Generator().manual_seed(5)).float()
        (out * G.to(DEV)).sum().backward()
        g2 = emb.emb2.weight.grad
        assert g2.is_sparse == sparse
        res[sparse] = (out.detach(), emb.emb1.weight.grad.clone(), g2.to_dense() if sparse else g2.clone())
    for a, b, what in zip(res[False], res[True], ("out", "g emb1", "g emb2")):
        assert torch.equal(a, b), what


@pytest.mark.parametrize("op", ["mult", "add", "cat"])
@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("D", [16, 12])
def test_qr_lookup_that_adds_the_field_offsets_itself(op, sparse, D):
    """QRHashingEmbedding(x, offsets=o) = (QRHashingEmbedding(x + o), x + o) — the model's `x + offsets`
    (src/models/dcn.py:204) inside the lookup kernel (mi_dual_gather_fwd_off); D = 12 (three float4 per row: no vector
    form) takes the fallback that adds first.  Same values and, on integer data, the same gradients bit for bit."""
    gen = torch.Generator().manual_seed(21)
    dims = [241, 8, 3697, 5, 31]
    B = 300
    off = torch.tensor([0] + dims[:-1]).cumsum(0)
    x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1)
    res = {}
    for inside in (False, True):
        gen = torch.Generator().manual_seed(22)
        emb = QRHashingEmbedding(dims, D, None, 2, op, sparse=sparse).to(DEV)
        with torch.no_grad():
            for w in (emb.emb1.weight, emb.emb2.weight):
                w.copy_(torch.randint(-3, 4, w.shape, generator=gen).float())
        if inside:
            assert emb.takes_offsets(x.to(DEV)) == (D == 16)
            out, rows = emb(x.to(DEV), offsets=off.to(DEV))
            assert torch.equal(rows.cpu(), x + off) and not rows.requires_grad
        else:
            out = emb((x + off).to(DEV))
        G = torch.randint(-2, 3, out.shape, generator=torch.Generator().manual_seed(5)).float()
        (out * G.to(DEV)).sum().backward()
        g2 = emb.emb2.weight.grad
        res[inside] = (out.detach(), emb.emb1.weight.grad.clone(), g2.to_dense() if g2.is_sparse else g2.clone())
    for a, b, what in zip(res[False], res[True], ("out", "g emb1", "g emb2")):
        assert torch.equal(a, b), what
    _lib.check_index_errors()


def test_qr_bag_modes():
    gen = torch.Generator().manual_seed(1)
    for mode in ("sum", "mean"):
        emb = QRHashingEmbedding([20, 30], 8, mode, 5, "mult").to(DEV)
        idx = torch.randint(0, 50, (7, 4), generator=gen)
        e1, e2 = emb.emb1.weight.detach().cpu(), emb.emb2.weight.detach().cpu()
        ref = torch.nn.functional.embedding_bag(idx % 5, e1, mode=mode) * torch.nn.functional.embedding_bag(idx // 5, e2, mode=mode)
        assert_close(emb(idx.to(DEV)), ref, 1e-6, 1e-6)


def test_qr_out_of_range_flagged():
    emb = QRHashingEmbedding(10, 8, None, 3, "add").to(DEV)
    out = emb(torch.tensor([1, 12, -1, 3], device=DEV))   # quotient 4 == emb2 rows -> out of range
    torch.cuda.synchronize()
    assert torch.count_nonzero(out[1]) == 0 and torch.count_nonzero(out[2]) == 0
    with pytest.raises(IndexError):
        _lib.check_index_errors()


# ------------------------------------------------------------------ CERP
@pytest.mark.parametrize("name", golden_names("cerp_"))
def test_cerp_matches_reference_golden(name):
    g = load_golden(name)
    emb = get_embedding({"name": "cerp", "bucket_size": int(g["bucket"])}, g["field_dims"].tolist(),
                        int(g["hidden"]), field_name="deepfm")
    assert emb.q_entity_per_row == int(g["q_entity_per_row"])
    emb.load_state_dict(g.group("param/"), strict=True)
    emb.to(DEV)
    for tag in ("1d", "2d"):
        emb.zero_grad()
        out = emb(g.t(f"x_{tag}").to(DEV))
        assert_close(out, g.t(f"out_{tag}"), 1e-6, 1e-7, f"out {tag}")   # sigmoid: expf vs torch CPU
        (out * g.t(f"G_{tag}").to(DEV)).sum().backward()
        for k in ("p_weight", "q_weight", "p_threshold", "q_threshold"):
            assert_close(getattr(emb, k).grad, g.t(f"grad_{tag}/{k}"), 1e-5, 1e-6, k)
    assert_close(emb.get_weight(), g.t("weight"), 1e-6, 1e-7, "get_weight")
    assert emb.get_num_params() == int(g["num_params"])


@pytest.mark.parametrize("N,D,bucket,shape", [(5000, 16, 100, (64, 26)), (37, 7, 5, (11,)), (200000, 64, 8000, (512,))])
def test_cerp_vs_oracle_with_pruning(N, D, bucket, shape):
    gen = torch.Generator().manual_seed(N)
    emb = CerpEmbedding(N, D, None, bucket).to(DEV)
    with torch.no_grad():
        emb.p_threshold.copy_(torch.randn(bucket, D, generator=gen) - 2.0)
        emb.q_threshold.copy_(torch.randn(bucket, D, generator=gen) - 2.0)
    idx = torch.randint(0, N, shape, generator=gen)
    out = emb(idx.to(DEV))
    p = {k: getattr(emb, k).detach().cpu().requires_grad_(True) for k in ("p_weight", "q_weight", "p_threshold", "q_threshold")}
    ref = ro.cerp_forward(idx, p["p_weight"], p["q_weight"], p["p_threshold"], p["q_threshold"], bucket, emb.q_entity_per_row)
    assert_close(out, ref, 1e-5, 1e-6, "forward")
    G = torch.randn(ref.shape, generator=gen)
    (ref * G).sum().backward()
    (out * G.to(DEV)).sum().backward()
    for k, v in p.items():
        assert_close(getattr(emb, k).grad, v.grad, 1e-4, 1e-5, k)


def test_cerp_retrain_from_checkpoint_dir(tmp_path):
    gen = torch.Generator().manual_seed(3)
    N, D, bucket = 300, 16, 20
    src = CerpEmbedding(N, D, None, bucket)
    with torch.no_grad():
        src.p_threshold.copy_(torch.randn(bucket, D, generator=gen) - 1.0)
        src.q_threshold.copy_(torch.randn(bucket, D, generator=gen) - 1.0)
    d = tmp_path / "deepfm"
    os.makedirs(d)
    torch.save(src.state_dict(), d / "initial.pth")
    torch.save(src.state_dict(), d / "target.pth")
    emb = RetrainCerpEmbedding(N, D, None, str(tmp_path), "deepfm", bucket_size=bucket).to(DEV)
    idx = torch.randint(0, N, (40, 3), generator=gen)
    out = emb(idx.to(DEV))
    pw = emb.p_weight.detach().cpu().requires_grad_(True)
    qw = emb.q_weight.detach().cpu().requires_grad_(True)
    ref = ro.cerp_retrain_forward(idx, pw, qw, emb.p_mask.cpu(), emb.q_mask.cpu(), bucket, emb.q_entity_per_row)
    assert_close(out, ref, 0, 0, "forward")
    G = torch.randn(ref.shape, generator=gen)
    (ref * G).sum().backward()
    (out * G.to(DEV)).sum().backward()
    assert_close(emb.p_weight.grad, pw.grad, 1e-5, 1e-6)
    assert_close(emb.q_weight.grad, qw.grad, 1e-5, 1e-6)
    assert emb.get_num_params() == int(emb.p_mask.sum() + emb.q_mask.sum())
    # sparse=True (torch.optim.SparseAdam's input, src/models/embeddings/cerp_embedding.py:219,230): same values, row form
    semb = RetrainCerpEmbedding(N, D, None, str(tmp_path), "deepfm", bucket_size=bucket, sparse=True).to(DEV)
    sout = semb(idx.to(DEV))
    assert_close(sout, ref, 0, 0, "forward (sparse=True)")
    (sout * G.to(DEV)).sum().backward()
    assert semb.p_weight.grad.is_sparse and semb.q_weight.grad.is_sparse
    assert_close(semb.p_weight.grad, pw.grad, 1e-5, 1e-6, "p grad (row form, coalesced)")
    assert_close(semb.q_weight.grad, qw.grad, 1e-5, 1e-6, "q grad (row form, coalesced)")


# ------------------------------------------------------------------ CSR-pruned
def test_pruned_embedding_golden():
    g = load_golden("csr_pruned")
    emb = PrunedEmbedding.from_weight(g.t("dense"))
    assert torch.equal(emb.values, g.t("values")) and torch.equal(emb.col_indices, g.t("col"))
    emb.to_cuda()
    assert emb.is_cuda
    assert torch.equal(emb(g.t("ids").to(DEV)).cpu(), g.t("out"))
    assert torch.equal(emb(g.t("ids2").to(DEV)).cpu(), g.t("out2"))
    assert torch.equal(emb.get_weight().cpu(), g.t("dense"))


def test_pruned_embedding_like_reference_test():
    # mirrors tests/test_emb.py:351-372 of the reference (90 %-sparse [1024,16] table)
    gen = torch.Generator().manual_seed(0)
    field_dims = [512, 512]
    num_items = sum(field_dims)
    model = pkg.VanillaEmbedding(field_dims, 16)
    nnz = int(num_items * 16 * 0.1)
    ind = torch.stack([torch.randint(num_items, (nnz,), generator=gen), torch.randint(16, (nnz,), generator=gen)])
    weight = torch.sparse_coo_tensor(ind, torch.ones(nnz), size=(num_items, 16)).to_dense()
    model._emb_module.weight.data = weight
    pruned = PrunedEmbedding.from_other_emb(model)
    model.to(DEV)
    pruned.to_cuda()
    inp = torch.randint(num_items, size=(256,), generator=gen).to(DEV)
    assert pruned(inp).isclose(model(inp)).all()
    assert pruned.get_weight().cpu().isclose(weight).all()
    big = torch.randint(num_items, size=(4096, 26), generator=gen)
    c = int_ops.csr_rows(pruned.values.cpu().numpy(), pruned.crow_indices.cpu().numpy(), pruned.col_indices.cpu().numpy(),
                         big.numpy(), 16)
    assert np.array_equal(pruned(big.to(DEV)).cpu().numpy(), c)


# ------------------------------------------------------------------ DHE
@pytest.mark.parametrize("name", golden_names("dhe_k"))
def test_dhe_matches_reference_golden(name):
    g = load_golden(name)
    DHEmbedding.COUNTER = int(g["prefix"])
    emb = DHEmbedding(g["field_dims"].tolist(), int(g["D"]), None, int(g["k"]), g["hidden"].tolist(), cached=False)
    DHEmbedding.COUNTER = 0
    # same seeded draws as the reference (torch CPU generator, seed 0) and same regenerated primes
    assert torch.equal(emb._slopes, g.t("slopes")) and torch.equal(emb._bias, g.t("bias"))
    assert torch.equal(emb._primes_choices, g.t("primes"))
    emb._seq.load_state_dict(g.group("param/_seq."), strict=True)
    emb.to(DEV).eval()
    for ids, ref in (("ids", "hash"), ("big_ids", "big_hash")):
        h = emb._get_universal_hash_batch(g.t(ids).to(DEV))
        assert torch.equal(h.cpu(), g.t(ref)), "hash features must be bit-exact"
    with torch.no_grad():
        assert_close(emb(g.t("ids").to(DEV)), g.t("out"), 1e-4, 1e-5, "uncached eval")
        assert_close(emb(g.t("x2").to(DEV)), g.t("out2"), 1e-4, 1e-5, "uncached eval 2d")
        emb._use_cache = True
        assert_close(emb(g.t("x2").to(DEV)), g.t("out2"), 1e-4, 1e-5, "on-the-fly == cached path")
        emb.compute_v2 = True
        assert_close(emb(g.t("x2").to(DEV)), g.t("out2"), 1e-4, 1e-5, "compute_v2")


@pytest.mark.parametrize("use_bn", [0, 1, 2])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("n,k,hidden,D", [(3000, 1024, [64], 16), (257, 64, [32, 32], 16), (70, 128, [], 8)])
def test_dhe_mlp_on_own_kernels_vs_float64_all_orderings(use_bn, training, n, k, hidden, D, monkeypatch):
    """DHEmbedding._forward_mlp (src/models/embeddings/dh_embedding.py:100-117,345-356) on mish_mlp.py's kernels for the
    three layer orderings (use_bn 0: Linear-Mish, 1: Linear-Mish-BatchNorm, 2: Linear-BatchNorm-Mish), training and eval:
    output, every parameter gradient and the BatchNorm running statistics against the same nn.Sequential in float64, held
    to a multiple of the stock float32 modules' own error; nn.Sequential.forward must not be what computes it."""
    import copy

    from recsys_benchmark_amd import mish_mlp as mm

    torch.manual_seed(n + k + use_bn)
    DHEmbedding.COUNTER = 0
    emb = DHEmbedding(n, D, None, k, list(hidden), use_bn=use_bn, cached=False)
    DHEmbedding.COUNTER = 0
    seq = emb._seq
    for m in seq:
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    seq.train(training)
    x = torch.rand(n, k) * 2 - 1
    G = torch.randn(n, D)
    refs = []
    for dt in (torch.float64, torch.float32):
        r = copy.deepcopy(seq).to(dt)
        out = r(x.to(dt))
        (out * G.to(dt)).sum().backward()
        refs.append((r, out))
    mine = copy.deepcopy(seq).to(DEV)
    xd = x.to(DEV)
    assert mm.mish_mlp_plan(mine, use_bn, xd) is not None
    monkeypatch.setattr(torch.nn.Sequential, "forward", lambda self, *a: (_ for _ in ()).throw(AssertionError("stock Sequential ran")))
    emb._seq = mine
    out = emb._forward_mlp(xd)
    (out * G.to(DEV)).sum().backward()
    monkeypatch.undo()

    def check(name, got, r64, r32, kf=16.0, floor=2e-6):
        got, r64, r32 = got.detach().double().cpu(), r64.detach().double(), r32.detach().double()
        scale = r64.abs().max().clamp_min(1e-30)
        err_f, err_s = (got - r64).abs().max() / scale, (r32 - r64).abs().max() / scale
        assert err_f <= max(kf * err_s, floor), f"{name}: own {err_f:.3e} vs stock float32 {err_s:.3e}"

    check("out", out, refs[0][1], refs[1][1])
    p64, p32, pm = dict(refs[0][0].named_parameters()), dict(refs[1][0].named_parameters()), dict(mine.named_parameters())
    for name in p64:
        if use_bn == 2 and training and name.endswith(".bias") and isinstance(seq[int(name.split(".")[0])], torch.nn.Linear):
            assert float(pm[name].grad.abs().max()) == 0.0      # a bias in front of a training-mode BatchNorm: exactly zero
            continue
        check(name, pm[name].grad, p64[name].grad, p32[name].grad)
    b64, b32, bm = dict(refs[0][0].named_buffers()), dict(refs[1][0].named_buffers()), dict(mine.named_buffers())
    for name in b64:
        if name.endswith("num_batches_tracked"):
            assert int(bm[name]) == int(b64[name])
        else:
            check(name, bm[name], b64[name], b32[name])
    with torch.no_grad():            # inference: same values, nothing kept
        out2 = emb._forward_mlp(xd) if not training else None
    if out2 is not None:
        assert torch.equal(out2, out.detach())


def test_dhe_per_item_hash_family_matches_reference_golden():
    """use_universal_hash=False (dh_embedding.py:155-196): the host-built feature table is bit-exact, the lookups are the
    HIP row gather + MLP, cached and per-unique-id flows agree with the reference's outputs."""
    g = load_golden("dhe_peritem")
    kw = dict(use_universal_hash=False)
    DHEmbedding.COUNTER = int(g["prefix"])
    emb = DHEmbedding(g["field_dims"].tolist(), int(g["D"]), None, int(g["k"]), g["hidden"].tolist(), cached=True, **kw)
    assert torch.equal(emb._cache, g.t("cache"))
    assert not any(k.startswith("_slopes") or k.startswith("_bias") for k in emb.state_dict())
    emb._seq.load_state_dict(g.group("param/_seq."), strict=True)
    emb.to(DEV).eval()
    DHEmbedding.COUNTER = int(g["prefix"])
    lazy = DHEmbedding(g["field_dims"].tolist(), int(g["D"]), None, int(g["k"]), g["hidden"].tolist(), cached=False, **kw)
    DHEmbedding.COUNTER = 0
    lazy._seq.load_state_dict(g.group("param/_seq."), strict=True)
    lazy.to(DEV).eval()
    with torch.no_grad():
        assert_close(emb(g.t("ids").to(DEV)), g.t("out"), 1e-4, 1e-5, "cached, 1-d ids")
        assert_close(emb(g.t("x2").to(DEV)), g.t("out2"), 1e-4, 1e-5, "cached, 2-d ids")
        assert_close(lazy(g.t("x2").to(DEV)), g.t("out_lazy"), 1e-4, 1e-5, "per unique id (cached=False, eval)")
        emb.compute_v2 = True
        assert_close(emb(g.t("x2").to(DEV)), g.t("out2"), 1e-4, 1e-5, "compute_v2")
    assert emb._cache.is_cuda, "the feature table lives on the MLP's device after the first lookup"
    lazy.train()
    with pytest.raises(NotImplementedError):
        lazy(g.t("x2").to(DEV))


def test_dhe_hash_full_scale_bit_exact():
    DHEmbedding.COUNTER = 12345
    emb = DHEmbedding(2000000000, 16, None, 1024, [64]).to(DEV)
    DHEmbedding.COUNTER = 0
    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(0, 2000000000, (4096,), generator=gen)
    h = emb._get_universal_hash_batch(ids.to(DEV)).cpu()
    _, f = int_ops.dhe_hash(ids.numpy(), emb._slopes.cpu().numpy(), emb._bias.cpu().numpy(),
                            emb._primes_choices.cpu().numpy(), 12345)
    assert np.array_equal(h.numpy(), f)
    assert h.min() >= -1 and h.max() <= 1


def test_dhe_counter_prefix_semantics():
    # tests/test_emb.py:64-109 of the reference: same COUNTER -> same features, else different
    c0 = DHEmbedding.COUNTER
    a = DHEmbedding(32, 64, None, 64, [64]).to(DEV)
    DHEmbedding.COUNTER = c0
    b = DHEmbedding(32, 64, None, 64, [64]).to(DEV)
    c = DHEmbedding(32, 64, None, 64, [64]).to(DEV)
    ids = torch.arange(32, device=DEV)
    assert torch.equal(a._get_universal_hash_batch(ids), b._get_universal_hash_batch(ids))
    assert not torch.equal(a._get_universal_hash_batch(ids), c._get_universal_hash_batch(ids))
    DHEmbedding.COUNTER = c0


# ------------------------------------------------------------------ DeepFM on compressed tables
@pytest.mark.parametrize("cfg", [
    {"name": "qr", "divider": 3, "operation": "mult"},
    {"name": "qr", "divider": 4, "operation": "cat"},
    {"name": "cerp", "bucket_size": 7},
])
def test_deepfm_with_compressed_embedding_vs_oracle(cfg):
    torch.manual_seed(5)
    dims, D, B = [5, 7, 11, 3], 8, 19
    m = pkg.DeepFM(dims, D, [16, 16], p_dropout=0.0, use_batchnorm=True, embedding_config=dict(cfg)).to(DEV)
    x = torch.stack([torch.randint(0, d, (B,)) for d in dims], 1)
    y = (torch.rand(B) < 0.4).float()
    logits = m(x.to(DEV))
    torch.nn.BCEWithLogitsLoss()(logits, y.to(DEV)).backward()
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    rows = x + p["offsets"]
    if cfg["name"] == "qr":
        emb = ro.qr_forward(rows, p["embedding.emb1.weight"], p["embedding.emb2.weight"], cfg["divider"], cfg["operation"])
    else:
        emb = ro.cerp_forward(rows, p["embedding.p_weight"], p["embedding.q_weight"], p["embedding.p_threshold"],
                              p["embedding.q_threshold"], 7, m.embedding.q_entity_per_row)
    y_fm = ro.first_order(rows, p["fc.weight"], p["_bias"]) + ro.fm_second_order(emb)
    ref = (y_fm + ro.mlp_tail(emb.reshape(B, -1), p, "_deep_branch", 2, True, True)).squeeze(-1)
    torch.nn.BCEWithLogitsLoss()(ref, y).backward()
    assert_close(logits, ref, 1e-4, 1e-5, "logits")
    named = dict(m.named_parameters())
    for k, v in p.items():
        if v.requires_grad and v.grad is not None and not k.startswith("linear_layer"):
            assert_close(named[k].grad, v.grad, 2e-4, 1e-5, f"grad {k}")


# ------------------------------------------------------------------ TT-Rec (torch semantics)
@pytest.mark.parametrize("name", golden_names("tt_"))
def test_tt_matches_reference_golden(name):
    from recsys_benchmark_amd.embeddings import TTRecTorch

    g = load_golden(name)
    ps, qs, rs = g["tt_p_shapes"].tolist(), g["tt_q_shapes"].tolist(), g["tt_ranks"].tolist()
    N, D = int(g["num_item"]), int(g["hidden"])
    explicit = name != "tt_r3x3"
    emb = TTRecTorch(N, D, rs[1:-1], tt_p_shapes=ps if explicit else None, tt_q_shapes=qs if explicit else None)
    assert emb.tt_p_shapes == ps and emb.tt_q_shapes == qs and emb.tt_ranks == rs   # incl. suggested_tt_shapes
    emb.load_state_dict(g.group("param/"), strict=True)
    emb.to(DEV)
    for tag in ("1d", "2d"):
        emb.zero_grad()
        out = emb(g.t(f"x_{tag}").to(DEV))
        assert_close(out, g.t(f"out_{tag}"), 1e-5, 1e-6, f"out {tag}")
        (out * g.t(f"G_{tag}").to(DEV)).sum().backward()
        for i in range(len(ps)):
            assert_close(emb.tt_cores[i].grad, g.t(f"grad_{tag}/tt_cores.{i}"), 1e-4, 1e-6, f"grad core {i}")
    assert_close(emb.get_weight(), g.t("weight"), 1e-5, 1e-6, "get_weight")
    assert emb.get_num_params() == sum(p * q * a * b for p, q, a, b in zip(ps, qs, rs[:-1], rs[1:]))


@pytest.mark.parametrize("name", golden_names("ttinit_"))
def test_tt_approx_uniform_construction_then_lookup(name):
    """weight_dist='approx-uniform' (tt_embedding_ops.py:861-986): seeded like the reference's set_seed, the three cores
    come out bit-identical; the HIP lookup of that table equals the reference's rows."""
    import random

    from recsys_benchmark_amd.embeddings import TTRecTorch

    g = load_golden(name)
    ps, qs, rs = g["tt_p_shapes"].tolist(), g["tt_q_shapes"].tolist(), g["tt_ranks"].tolist()
    np.random.seed(2023), torch.manual_seed(2023), random.seed(2023)
    explicit = name.endswith("r2x3")
    emb = TTRecTorch(int(g["num_item"]), int(g["hidden"]), rs[1:-1], tt_p_shapes=ps if explicit else None,
                     tt_q_shapes=qs if explicit else None, weight_dist="approx-uniform")
    for i, c in enumerate(emb.tt_cores):
        assert torch.equal(c.data, g.t(f"param/tt_cores.{i}")), f"core {i}"
    emb.to(DEV)
    assert_close(emb(g.t("x").to(DEV)), g.t("out"), 1e-5, 1e-6, "lookup")
    assert_close(emb.get_weight(), g.t("weight"), 1e-5, 1e-6, "get_weight")


def test_tt_reference_ranks_vs_oracle():
    # the reference's DeepFM config: tt_ranks [128, 96] (configs/deepfm/tt_rec.yaml:10)
    from recsys_benchmark_amd.embeddings import TTRecTorch

    gen = torch.Generator().manual_seed(4)
    emb = TTRecTorch(20000, 16, [128, 96], tt_p_shapes=[25, 25, 32], tt_q_shapes=[2, 2, 4], weight_dist="normal")
    with torch.no_grad():
        for c in emb.tt_cores:
            c.copy_(torch.randn(c.shape, generator=gen) * 0.1)
    idx = torch.randint(0, 20000, (64, 3), generator=gen)
    cores = [c.detach().clone().requires_grad_(True) for c in emb.tt_cores]
    ref = ro.tt_forward(idx.flatten(), emb.tt_p_shapes, emb.tt_q_shapes, emb.tt_ranks, cores).reshape(64, 3, 16)
    G = torch.randn(ref.shape, generator=gen)
    (ref * G).sum().backward()
    emb.to(DEV)
    out = emb(idx.to(DEV))
    assert_close(out, ref, 1e-4, 1e-5, "forward")
    (out * G.to(DEV)).sum().backward()
    for i, c in enumerate(cores):
        assert_close(emb.tt_cores[i].grad, c.grad, 1e-3, 1e-4, f"grad core {i}")
    emb(torch.tensor([20000, -1], device=DEV))
    with pytest.raises(IndexError):
        _lib.check_index_errors()


# ------------------------------------------------------------------ PEP / PEP-retrain (§8f rank 4)
@pytest.mark.parametrize("ttype", ["global", "dimension", "feature", "feature_dim"])
def test_pep_matches_reference_golden(ttype, tmp_path):
    g = load_golden(f"pep_{ttype}")
    emb = get_embedding({"name": "pep", "threshold_type": ttype, "checkpoint_weight_dir": str(tmp_path)},
                        g["field_dims"].tolist(), int(g["hidden"]), field_name="deepfm")
    emb.load_state_dict(g.group("param/"), strict=True)
    emb.to(DEV)
    for tag in ("1d", "2d"):
        emb.zero_grad()
        out = emb(g.t(f"x_{tag}").to(DEV))
        assert_close(out, g.t(f"out_{tag}"), 1e-6, 1e-7, f"out {tag}")
        (out * g.t(f"G_{tag}").to(DEV)).sum().backward()
        assert_close(emb.emb.weight.grad, g.t(f"grad_{tag}/emb.weight"), 1e-5, 1e-6, "g weight")
        assert_close(emb.s.grad, g.t(f"grad_{tag}/s"), 1e-4, 1e-6, "g s")
    assert_close(emb.get_weight(), g.t("weight"), 1e-6, 1e-7, "get_weight")
    assert emb.get_num_params() == int(g["num_params"])


def test_pep_retrain_matches_reference_golden(tmp_path):
    g = load_golden("pep_retrain")
    d = tmp_path / "deepfm"
    os.makedirs(d)
    torch.save({"emb.weight": g.t("ckpt_weight"), "s": g.t("ckpt_s")}, d / "0.5.pth")
    emb = get_embedding({"name": "pep_retrain", "checkpoint_weight_dir": str(tmp_path), "sparsity": 0.5},
                        g["field_dims"].tolist(), int(g["hidden"]), field_name="deepfm")
    emb.load_state_dict(g.group("param/"), strict=True)
    assert int(emb.get_num_params()) == int(g["nnz"])
    emb.to(DEV)
    out = emb(g.t("x").to(DEV))
    assert_close(out, g.t("out"), 0, 0, "out")
    (out * g.t("G").to(DEV)).sum().backward()
    assert_close(emb.emb.weight.grad, g.t("grad/emb.weight"), 1e-5, 1e-6)
    assert_close(emb.get_weight(), g.t("weight"), 0, 0)
    # sparse=True (src/models/embeddings/pep_embedding.py:215-221 with sparse=self._sparse): the same gradient in row form
    semb = get_embedding({"name": "pep_retrain", "checkpoint_weight_dir": str(tmp_path), "sparsity": 0.5, "sparse": True},
                         g["field_dims"].tolist(), int(g["hidden"]), field_name="deepfm")
    semb.load_state_dict(g.group("param/"), strict=True)
    semb.to(DEV)
    sout = semb(g.t("x").to(DEV))
    assert_close(sout, g.t("out"), 0, 0, "out (sparse=True)")
    (sout * g.t("G").to(DEV)).sum().backward()
    assert semb.emb.weight.grad.is_sparse
    assert_close(semb.emb.weight.grad, g.t("grad/emb.weight"), 1e-5, 1e-6, "row-form grad, coalesced")


@pytest.mark.parametrize("ttype", ["global", "dimension", "feature", "feature_dim"])
def test_pep_vs_oracle_large(ttype, tmp_path):
    gen = torch.Generator().manual_seed(21)
    N, D = 50000, 16
    emb = get_embedding({"name": "pep", "threshold_type": ttype, "checkpoint_weight_dir": str(tmp_path)}, N, D).to(DEV)
    with torch.no_grad():
        emb.s.copy_((torch.randn(emb.s.shape, generator=gen) * 0.5 - 4.0).to(DEV))   # sigmoid ~ 0.018 vs xavier bound 0.011..
        emb.emb.weight.mul_(5.0)
    idx = torch.randint(0, N, (4096, 26), generator=gen)
    out = emb(idx.to(DEV))
    w = emb.emb.weight.detach().cpu().requires_grad_(True)
    s = emb.s.detach().cpu().requires_grad_(True)
    ref = ro.pep_forward(idx, w, s)
    assert_close(out, ref, 1e-5, 1e-7, "forward")
    G = torch.randn(ref.shape, generator=gen)
    (ref * G).sum().backward()
    (out * G.to(DEV)).sum().backward()
    assert_close(emb.emb.weight.grad, w.grad, 1e-4, 1e-5, "g weight")
    sc = max(1.0, float(s.grad.abs().max()))
    assert_close(emb.s.grad, s.grad, 1e-3, 1e-4 * sc, "g s")


# ------------------------------------------------------------------ PTQ tables (§8f rank 4)
def test_ptq_matches_reference_golden(tmp_path):
    from recsys_benchmark_amd.embeddings.ptq_emb import PTQEmb_Fp16, PTQEmb_Int

    g = load_golden("ptq")
    path = str(tmp_path / "c.pth")
    torch.save({"state_dict": {"embedding._emb_module.weight": g.t("W")}}, path)
    x = g.t("x").to(DEV)
    e = PTQEmb_Fp16(None, None, None, path)
    assert torch.equal(e.weight.view(torch.int16), g.t("fp16_weight"))
    assert torch.equal(e.to(DEV)(x).cpu(), g.t("fp16_out")), "fp16 -> fp32 is exact"
    for bits in (8, 16):
        e = PTQEmb_Int(None, None, None, path, n_bits=bits)
        assert torch.equal(e.weight, g.t(f"int{bits}_weight")), "integer codes must be bit-exact"
        assert torch.equal(e.bias, g.t(f"int{bits}_bias")) and torch.equal(e.scale, g.t(f"int{bits}_scale"))
        assert_close(e.get_weight(), g.t(f"int{bits}_full"), 0, 0)
        assert_close(e.to(DEV)(x), g.t(f"int{bits}_out"), 0, 0, f"int{bits} dequantised rows")


@pytest.mark.parametrize("n_bits", [16, 8])
def test_ptq_known_codes(n_bits, tmp_path):      # tests/test_emb.py:529-561 of the reference
    from recsys_benchmark_amd.embeddings.ptq_emb import PTQEmb_Int

    bias, scale = 0, 0.1
    q_min, q_max = (-1) * (1 << (n_bits - 1)), (1 << (n_bits - 1)) - 1
    r_min = (q_min - bias) * scale
    r_max = scale * (q_max - q_min) + r_min
    w = torch.tensor([[r_min, 0.11, 0.22, 0.31, 0.66, 0.82, 1.31, r_max]], dtype=torch.float32)
    codes = torch.tensor([[q_min, 1, 2, 3, 7, 8, 13, q_max]], dtype=torch.int16)
    path = str(tmp_path / "t.pth")
    torch.save({"state_dict": {"embedding._emb_module.weight": w}}, path)
    emb = PTQEmb_Int([1], 8, None, path, n_bits)
    assert (emb.weight == codes).all()
    out = emb.to(DEV)(torch.zeros(3, dtype=torch.long, device=DEV))
    assert out.cpu().isclose((codes * scale).expand(3, 8)).all()


# ------------------------------------------------------------------ QAT (stochastic rounding)
@pytest.mark.parametrize("name", ["qat_int8", "qat_int16"])
def test_qat_matches_reference_golden(name):
    g = load_golden(name)
    bits = int(g["n_bits"])
    emb = get_embedding({"name": "qat", "n_bits": bits}, g["field_dims"].tolist(), int(g["hidden"]))
    assert set(emb.state_dict()) == {"_emb_module.weight", "scale"}
    emb.load_state_dict({"_emb_module.weight": g.t("param/_emb_module.weight"), "scale": g.t("param/scale")})
    emb.to(DEV)
    out = emb(g.t("x").to(DEV), prob=g.t("prob").to(DEV))
    assert_close(out, g.t("out"), 0, 0, "rounded rows (same draw): bit-exact")
    (out * g.t("G").to(DEV)).sum().backward()
    assert_close(emb._emb_module.weight.grad, g.t("grad/_emb_module.weight"), 1e-6, 1e-7, "row gradient")
    # a sum of ~150 products of magnitude up to 127 * |G|: fp32 order
    assert_close(emb.scale.grad, g.t("grad/scale"), 1e-5, 1e-4, "scale gradient")
    assert torch.equal(emb.get_weight(), emb._emb_module.weight)


def test_qat_initial_scale_and_rounding_statistics():
    """scale init (qat_emb.py:111-115); with the library's own generator every output is one of the two grid
    neighbours of w/scale, rounds up with probability frac(w/scale) (unbiased), exact values never move, and
    the backward sees the same draw as the forward."""
    torch.manual_seed(0)
    emb = get_embedding({"name": "qat", "n_bits": 8}, [50, 30], 16).to(DEV)
    W = emb._emb_module.weight.detach().clone()
    assert_close(emb.scale.detach(), (W.max() - W.min()) / 255.0, 1e-6, 0, "scale init")
    s = float(emb.scale.detach())
    x = torch.randint(0, 80, (4096, 4), device=DEV)
    q = W[x] / s
    outs = torch.stack([emb(x).detach() for _ in range(64)])
    lo, hi = torch.floor(q) * s, (torch.floor(q) + 1) * s
    assert bool(((outs == lo) | (outs == hi)).all()), "only the two neighbouring grid points"
    p_up = (outs == hi).float().mean(0)
    assert float((p_up - (q - torch.floor(q))).abs().mean()) < 0.05          # 64 draws: sigma ~ 0.06 per element
    assert abs(float((outs.mean(0) - W[x]).mean())) < 2e-3 * s, "unbiased on average"
    assert float((outs[0] != outs[1]).float().mean()) > 0.2, "a fresh stream every forward"
    with torch.no_grad():
        emb._emb_module.weight.copy_(torch.round(W / s) * s)                  # already on the grid
    on_grid = emb(x).detach()
    assert_close(on_grid, emb._emb_module.weight.detach()[x], 1e-6, 1e-7, "grid points are fixed points")
    # backward re-derives the forward's rounding: dscale = sum g * (rounded - w/scale) inside the clamps
    with torch.no_grad():
        emb._emb_module.weight.copy_(W)
    out = emb(x)
    G = torch.randn_like(out)
    (out * G).sum().backward()
    qf = W[x] / s
    m = torch.where(qf >= 127, torch.full_like(qf, 127.0), torch.where(qf <= -128, torch.full_like(qf, -128.0),
                                                                     out.detach() / s - qf))
    want = (G * m).sum()
    assert_close(emb.scale.grad, want, 1e-3, 1e-2, "scale gradient from the re-derived rounding")
    dense = torch.zeros_like(W).index_add_(0, x.view(-1), G.view(-1, 16))
    assert_close(emb._emb_module.weight.grad, dense, 1e-4, 1e-5, "straight-through rows")


def test_qat_bag_mode_rounds_the_reduced_bag():
    torch.manual_seed(1)
    emb = get_embedding({"name": "qat", "n_bits": 16}, [40], 8, mode="sum").to(DEV)
    x = torch.randint(0, 40, (9, 3), device=DEV)
    prob = torch.rand(9, 8, device=DEV)
    out = emb(x, prob=prob)
    W, s = emb._emb_module.weight.detach().cpu(), emb.scale.detach().cpu()
    ref = ro.qat_forward(torch.arange(9), W[x.cpu()].sum(1), s, 16, prob.cpu())
    assert_close(out, ref, 1e-6, 1e-7, "bag then round")


# ------------------------------------------------------------------ OptEmbed supernet lookup (DeepFM variant)
@pytest.mark.parametrize("name", ["optembed_l1", "optembed_l2"])
def test_optembed_matches_reference_golden(name):
    g = load_golden(name)
    dims, D, norm = g["field_dims"].tolist(), int(g["hidden"]), int(g["norm"])
    emb = get_embedding({"name": "deepfm_optembed", "t_init": 0.0, "norm": norm}, dims, D)
    want = {"_weight", "_mask_e_module._t_param", "_mask_e_module._field_dims", "_full_mask_d"}
    assert set(emb.state_dict()) == want
    emb.load_state_dict({k: g.t("param/" + k) for k in want})
    emb.to(DEV).train()
    emb._forced_mask_d = g.t("mask_d").to(DEV)            # the reference's torch.randint draw
    out = emb(g.t("x").to(DEV))
    assert_close(out, g.t("out"), 0, 0, "masked rows: exact copies or zeros")
    (out * g.t("G").to(DEV)).sum().backward()
    # straight-through terms: a wave reduction of D products, scattered with float atomics
    assert_close(emb._weight.grad, g.t("grad/_weight"), 1e-5, 1e-6, "table gradient")
    assert_close(emb._mask_e_module._t_param.grad, g.t("grad/_mask_e_module._t_param"), 1e-5, 1e-6, "threshold gradient")
    assert_close(emb.get_l_s(), g.t("l_s"), 1e-6, 0)
    del emb._forced_mask_d
    emb.eval()
    assert_close(emb.get_weight(), g.t("weight_eval"), 0, 0, "row-masked table")
    assert_close(emb.get_weight(g.t("mask_d_field").to(DEV)), g.t("weight_eval_masked"), 0, 0, "row + dimension masks")
    assert_close(emb(g.t("x").to(DEV)), g.t("lookup_eval_masked"), 0, 0, "eval lookups reuse the mask given to get_weight")
    assert emb.get_num_params() == int(g["nnz"])
    sub = emb.get_submask()
    alive = (g.t("weight_eval").abs().sum(1) > 0).long()
    assert sub.tolist() == [int(alive[a:b].sum()) for a, b in ((0, 13), (13, 42), (42, 49))]


def test_optembed_feature_thresholds_and_own_sampling():
    g = load_golden("optembed_feature")
    dims, D = g["field_dims"].tolist(), int(g["hidden"])
    emb = get_embedding({"name": "deepfm_optembed", "t_init": 0.0, "mode_threshold_e": "feature",
                         "mode_threshold_d": "feature"}, dims, D)
    emb.load_state_dict({k: g.t("param/" + k) for k in emb.state_dict()})
    emb.to(DEV).eval()
    assert_close(emb.get_weight(), g.t("weight_eval"), 0, 0)
    assert_close(emb.get_weight(g.t("mask_d").to(DEV)), g.t("weight_eval_masked"), 0, 0)
    assert_close(emb(g.t("x").to(DEV)), g.t("lookup_eval_masked"), 0, 0)
    # 'deepfm_optembed_d': no row mask; training draws a fresh prefix mask per (sample, field)
    torch.manual_seed(0)
    d_only = get_embedding({"name": "deepfm_optembed_d"}, dims, D).to(DEV).train()
    assert "_mask_e_module._t_param" not in d_only.state_dict()
    x = g.t("x").to(DEV)
    a, b = d_only(x).detach(), d_only(x).detach()
    full = d_only._weight.detach()[x]
    kept = a != 0
    assert bool((a[kept] == full[kept]).all()) and bool(kept[..., 0].all()), "kept entries are the table's, dim 0 always kept"
    assert bool((kept[..., :-1] >= kept[..., 1:]).all()), "prefix masks"
    assert not torch.equal(a, b), "a new draw every forward"


@pytest.mark.parametrize("op", ["mult", "add"])
@pytest.mark.parametrize("div2", [2, 7, 2 ** 33])
@pytest.mark.parametrize("De", [16, 24, 4])
def test_dual_gather_gradients_with_small_and_huge_ids(op, div2, De):
    """mi_dual_gather_fwd / _bwd against torch indexing / index_add on integer-valued data (exact), with ids below and far
    beyond 2^32 (div2 = 2^33) and row widths that do and do not divide a wave."""
    gen = torch.Generator().manual_seed(De + (div2 % 1000))
    n1, n2, n = 5, 9, 3000
    i2 = torch.randint(0, n2, (n,), generator=gen)
    idx = i2 * div2 + torch.randint(0, min(div2, 1000), (n,), generator=gen)      # anywhere inside bucket i2
    i1 = idx % n1
    mk = lambda *s: torch.randint(-3, 4, s, generator=gen).float()          # noqa: E731
    T1, T2, G = mk(n1, De), mk(n2, De), mk(n, De)
    a, b = T1.to(DEV).requires_grad_(True), T2.to(DEV).requires_grad_(True)
    out = _kernels.dual_gather(idx.to(DEV), a, b, n1, div2, op)
    ref = T1[i1] * T2[i2] if op == "mult" else T1[i1] + T2[i2]
    assert torch.equal(out.cpu(), ref)
    (out * G.to(DEV)).sum().backward()
    g1 = torch.zeros(n1, De).index_add_(0, i1, G * (T2[i2] if op == "mult" else 1.0))
    g2 = torch.zeros(n2, De).index_add_(0, i2, G * (T1[i1] if op == "mult" else 1.0))
    assert torch.equal(a.grad.cpu(), g1) and torch.equal(b.grad.cpu(), g2)


@pytest.mark.parametrize("op", ["mult", "add", "cat"])
def test_dual_gather_backward_with_the_small_field_hint_is_the_same_gradient(op):
    """Fields with a handful of values (here 3, 5, 9, 31 and 32 of them among larger ones, odd offsets so that spans straddle
    quotient rows, plus ids OUTSIDE their field's range in one column) get their table-2 gradient summed per field in
    registers (mi_dual_gather_bwd_fields) instead of by same-address atomics: integer-valued data, so both are exact."""
    dims = [3, 700, 5, 9, 1201, 31, 32, 64, 2, 4000]
    gen = torch.Generator().manual_seed(5)
    B, D, div = 513, 16, 2
    off = torch.tensor([0] + dims[:-1]).cumsum(0)
    ids = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1) + off
    ids[:7, 0] = torch.randint(0, sum(dims), (7,), generator=gen)          # ids that are not in field 0's range
    N = sum(dims)
    De = D // 2 if op == "cat" else D
    T1v = torch.randint(-3, 4, (div, De), generator=gen).float()
    T2v = torch.randint(-3, 4, ((N + div - 1) // div, De), generator=gen).float()
    G = torch.randint(-2, 3, (B, 2 * len(dims), De) if op == "cat" else (B, len(dims), De), generator=gen).float().to(DEV)
    hint = _kernels.small_field_hint(dims, div, DEV)
    assert hint is not None and hint[0].tolist() == [0, 2, 3, 5, 8]        # (32 values at an odd offset span 17 rows: not small)
    grads = []
    for fields in (None, hint):
        T1 = T1v.clone().to(DEV).requires_grad_(True)
        T2 = T2v.clone().to(DEV).requires_grad_(True)
        out = _kernels.dual_gather(ids.to(DEV), T1, T2, mod1=div, div2=div, op=op, fields=fields)
        out.backward(G)
        grads.append((T1.grad.clone(), T2.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
