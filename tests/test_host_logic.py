"""CPU: host-side logic of the drop-in — registry, constructor conventions, table sizing,
state_dict keys and checkpoint helpers — against what the reference exposes."""
import copy

import pytest
import torch

from conftest import load_golden

import recsys_benchmark_amd as pkg
from recsys_benchmark_amd.embeddings import NAME_TO_CLS, OUT_OF_SCOPE, detect_special, get_embedding
from recsys_benchmark_amd.embeddings.dh_embedding import DHEmbedding, large_primes


def test_registry_keys_cover_the_reference_registry():
    reference_keys = {"vanilla", "qr", "dhe", "pep", "pep_retrain", "optembed_d", "optembed_d_retrain", "optembed",
                      "optembed_retrain", "deepfm_optembed", "deepfm_optembed_d", "deepfm_optembed_retrain", "tt_emb",
                      "tt_emb_torch", "cerp", "cerp_retrain", "qat"}
    assert reference_keys <= set(NAME_TO_CLS) | set(OUT_OF_SCOPE) | {"tt_emb_torch"}
    for k in OUT_OF_SCOPE:
        with pytest.raises(NotImplementedError):
            get_embedding({"name": k}, [3, 4], 8)
    assert {"vanilla", "qr", "dhe", "cerp", "cerp_retrain", "tt_emb_torch", "pep", "pep_retrain"} <= set(NAME_TO_CLS)
    with pytest.raises(NotImplementedError):
        get_embedding({"name": "nope"}, [3, 4], 8)


def test_get_embedding_does_not_mutate_config_and_forwards_field_name():
    cfg = {"name": "cerp", "bucket_size": 5}
    before = copy.deepcopy(cfg)
    emb = get_embedding(cfg, [3, 4], 8, field_name="deepfm")
    assert cfg == before and emb.field_name == "deepfm"
    with pytest.raises(AssertionError):
        get_embedding({"name": "vanilla"}, 3, 8, mode="prod")


@pytest.mark.parametrize("num_item,divider,expected", [(32, 8, 4), (31, 8, 4), (33, 8, 5), (1, 8, 1)])
def test_qr_table_sizing(num_item, divider, expected):      # tests/test_emb.py:51-61 of the reference
    emb = get_embedding({"name": "qr", "divider": divider}, num_item, 16)
    assert emb.emb2.num_embeddings == expected and emb.emb1.num_embeddings == divider


def test_state_dict_keys_match_reference_goldens():
    g = load_golden("deepfm_small_bn_train")
    m = pkg.DeepFM([5, 7, 11], 4, [8, 8], p_dropout=0.0, use_batchnorm=True)
    assert set(m.state_dict()) == set(g.group("param/"))
    g = load_golden("lightgcn_L2")
    assert set(pkg.LightGCN(77, 102, 2, 16).state_dict()) == set(g.group("param/"))
    g = load_golden("single_lightgcn_L2")
    assert set(pkg.SingleLightGCN(77, 102, 2, 16).state_dict()) == set(g.group("param/"))
    g = load_golden("qr_mult_div2")
    assert set(get_embedding({"name": "qr", "divider": 2}, [13, 29, 7], 8).state_dict()) == set(g.group("param/"))
    g = load_golden("cerp_default")
    assert set(get_embedding({"name": "cerp", "bucket_size": 10}, [13, 29, 7], 8).state_dict()) == set(g.group("param/"))


def test_deepfm_offsets_and_load(tmp_path):
    dims = [5, 7, 11]
    m = pkg.get_ctr_model(dims, {"name": "deepfm", "num_factor": 4, "hidden_sizes": [8]})
    assert m.offsets.tolist() == [[0, 5, 12]]
    ckpt = {"state_dict": m.state_dict(), "model_config": {"num_factor": 4, "hidden_sizes": [8]}, "field_dims": dims}
    path = tmp_path / "c.pth"
    torch.save(ckpt, path)
    m2 = pkg.DeepFM.load(str(path))
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    pkg.save_ctr_checkpoint(m, str(tmp_path))
    assert (tmp_path / "deepfm" / "target.pth").exists()


def test_packed_tables_save_the_references_checkpoint_layout(tmp_path):
    """DeepFM.pack_tables() turns the two lookup tables into column views of one [N, 32] buffer; every way of asking for a
    state_dict (the model's, and `model.embedding`'s as save_ctr_checkpoint does) must still give the reference's two
    contiguous tensors, not views that serialise the whole packed storage behind them."""
    import os

    dims = [50, 70, 130]
    m = pkg.get_ctr_model(dims, {"name": "deepfm", "num_factor": 16, "hidden_sizes": [8]})
    W0, w0 = m.embedding._emb_module.weight.detach().clone(), m.fc.weight.detach().clone()
    m.pack_tables()
    assert m.tables_packed and not m.embedding._emb_module.weight.is_contiguous()
    N = sum(dims)
    for sd in (m.state_dict(), m.embedding.state_dict(), m.embedding._emb_module.state_dict(), m.fc.state_dict()):
        for k, t in sd.items():
            assert t.is_contiguous(), k
            assert t.untyped_storage().nbytes() == t.numel() * t.element_size(), f"{k} drags a larger storage along"
    pkg.save_ctr_checkpoint(m, str(tmp_path))
    path = tmp_path / "deepfm" / "target.pth"
    assert os.path.getsize(path) < N * 16 * 4 + 4096, "the file holds the [N, 16] table, not the [N, 32] packed buffer"
    sd = torch.load(path)
    assert list(sd) == ["_emb_module.weight"] and sd["_emb_module.weight"].is_contiguous()
    assert torch.equal(sd["_emb_module.weight"], W0)
    full = m.state_dict()
    assert torch.equal(full["fc.weight"], w0) and torch.equal(full["embedding._emb_module.weight"], W0)


def test_graph_model_factory_and_checkpoint_helpers(tmp_path):
    cfg = {"name": "lightgcn", "num_layers": 3, "hidden_size": 8}
    m = pkg.get_graph_model(10, 12, cfg)
    assert cfg["name"] == "lightgcn" and isinstance(m, pkg.LightGCN) and m.num_layers == 3
    assert [n for n, _ in m.get_embs()] == ["user", "item"]
    pkg.save_cf_emb_checkpoint(m, str(tmp_path))
    assert (tmp_path / "user" / "target.pth").exists() and (tmp_path / "item" / "target.pth").exists()
    torch.save({"state_dict": m.state_dict(), "model_config": dict(cfg), "num_users": 10, "num_items": 12}, tmp_path / "g.pth")
    m2 = pkg.load_graph_model(str(tmp_path / "g.pth"))
    assert torch.equal(m2.user_emb_table.get_weight(), m.user_emb_table.get_weight())
    with pytest.raises(NotImplementedError):
        pkg.get_graph_model(3, 3, {"name": "hccf"})


def test_detect_special():
    assert detect_special({"model": {"embedding_config": {"name": "cerp_retrain"}}}) == ("cerp", True)
    assert detect_special({"model": {}}) == (None, False)
    assert detect_special({"model": {"embedding_config": {"name": "optembed_d"}}}) == ("optembed_d", False)


def test_dhe_seeded_constants_and_counter():
    assert len(large_primes()) == 74518 and int(large_primes()[0]) == 1000003 and int(large_primes()[-1]) == 2059181
    g = load_golden("dhe_k64")
    DHEmbedding.COUNTER = int(g["prefix"])
    emb = DHEmbedding(g["field_dims"].tolist(), 16, None, 64, [32, 32])
    assert emb._prefix == int(g["prefix"]) and DHEmbedding.COUNTER == int(g["prefix"]) + 54
    DHEmbedding.COUNTER = 0
    assert torch.equal(emb._slopes, g.t("slopes")) and torch.equal(emb._primes_choices, g.t("primes"))
    assert emb.get_extra_state() == {"_prefix": int(g["prefix"])}


def test_cerp_threshold_init_and_entity_per_row():
    emb = get_embedding({"name": "cerp", "bucket_size": 10}, [13, 29, 7], 8, field_name="x")
    assert emb.q_entity_per_row == 5 and torch.all(emb.q_threshold == -100.0)


# ------------------------------------------------------------------ sharded lookup: host-side pieces
def test_bucket_capacity_and_row_ownership():
    from recsys_benchmark_amd.sharded import bucket_capacity, local_num_rows

    assert bucket_capacity(1000, 1, 1.25) == 1000                       # one rank: every lookup stays home
    n = 4096 * 26
    for world in (2, 4, 8):
        cap = bucket_capacity(n, world, 1.25)
        mean = -(-n // world)
        assert mean < cap <= n and cap >= int(mean * 1.25)
        assert bucket_capacity(n, world, float(world)) == n             # slack = world can never overflow
    assert sum(local_num_rows(33_762_577, r, 8) for r in range(8)) == 33_762_577


def test_routing_restatement_properties():
    """The torch restatement the HIP router is checked against (oracle/sharded_ops.py): every lookup gets a unique
    slot in its owner's bucket in lookup order; unused slots address the owner's sink row; overflow and
    out-of-range lookups land on the dump slot."""
    from oracle.sharded_ops import TorchOps

    gen = torch.Generator().manual_seed(0)
    world, N, cap = 4, 1003, 90
    x = torch.randint(0, N, (60, 5), generator=gen)
    x[3, 2], x[7, 0] = N + 5, -2
    of = torch.zeros(1, dtype=torch.int32)
    send, slot = TorchOps.route_buckets(x, None, world, N, cap, of)
    flat, s = x.view(-1), slot.view(-1)
    ok = (flat >= 0) & (flat < N)
    assert bool((s[~ok] == world * cap).all()) and int(of) == 0
    live = s[ok]
    assert live.unique().numel() == live.numel() and bool((live // cap == flat[ok] % world).all())
    assert torch.equal(send[live], flat[ok] // world)
    for w in range(world):
        mine = live[live // cap == w]
        assert torch.equal(mine, torch.sort(mine)[0]) and int(mine.min()) == w * cap    # lookup order, packed from the left
        pad = torch.ones(cap, dtype=torch.bool)
        pad[mine - w * cap] = False
        assert bool((send[w * cap:(w + 1) * cap][pad] == (N - w + world - 1) // world).all())   # sink row
    of2 = torch.zeros(1, dtype=torch.int32)
    _, slot2 = TorchOps.route_buckets(torch.zeros(50, 1, dtype=torch.int64), None, world, N, 8, of2)
    assert int(of2) == 1 and int((slot2 == world * 8).sum()) == 42


def test_tt_grouped_path_selection():
    from recsys_benchmark_amd import _kernels

    assert _kernels.tt_grouped_supported(100_000, [2, 2, 4], [1, 128, 96, 1])
    assert not _kernels.tt_grouped_supported(100, [2, 2, 4], [1, 128, 96, 1])            # too few lookups
    assert not _kernels.tt_grouped_supported(100_000, [16], [1, 1])                      # a single core is a plain table
    assert not _kernels.tt_grouped_supported(100_000, [4, 4], [1, 6, 1])                 # rank not a multiple of 4
    assert _kernels._tt_last_fast(4, 96, 4) and _kernels._tt_last_fast(2, 64, 8)
    assert not _kernels._tt_last_fast(4, 96, 3)                                          # 12 outputs do not tile a wave
    assert not _kernels._tt_last_fast(8, 128, 4)                                         # chunk of 1024 floats > 512


def test_registry_forced_arguments_and_field_name():
    from recsys_benchmark_amd.embeddings import _FORCED, _WANTS_FIELD_NAME

    assert _FORCED["deepfm_optembed_d"] == {"t_init": None} and _FORCED["deepfm_optembed"] == {}
    assert _WANTS_FIELD_NAME == {"pep", "pep_retrain", "cerp", "cerp_retrain"}
    cfg = {"name": "qr", "divider": 3}
    import recsys_benchmark_amd.embeddings as E
    emb = E.get_embedding(cfg, [5, 7], 8)
    assert cfg == {"name": "qr", "divider": 3} and type(emb).__name__ == "QRHashingEmbedding"


def test_binary_auc_is_sklearns_roc_auc_with_ties():
    from sklearn.metrics import roc_auc_score

    from recsys_benchmark_amd.trainer import binary_auc

    gen = torch.Generator().manual_seed(0)
    for n, levels in ((1000, 0), (500, 7), (64, 2)):
        score = torch.rand(n, generator=gen)
        if levels:
            score = (score * levels).floor() / levels          # many tied scores
        true = (torch.rand(n, generator=gen) < 0.4).float()
        assert abs(binary_auc(true, score) - roc_auc_score(true.tolist(), score.tolist())) < 1e-12
    with pytest.raises(ValueError):
        binary_auc(torch.ones(5), torch.rand(5))


def test_adam_subclass_on_cpu_parameters_is_plain_torch_adam():
    """optim.Adam only switches to the HIP kernel (and device-side step counts) for GPU parameters."""
    from recsys_benchmark_amd.optim import Adam

    torch.manual_seed(0)
    p, q = torch.nn.Parameter(torch.randn(7, 3)), None
    q = torch.nn.Parameter(p.detach().clone())
    a, b = Adam([p], lr=1e-2, weight_decay=1e-3), torch.optim.Adam([q], lr=1e-2, weight_decay=1e-3)
    assert a.param_groups[0]["capturable"] is False
    for _ in range(3):
        g = torch.randn(7, 3)
        p.grad, q.grad = g.clone(), g.clone()
        a.step()
        b.step()
    assert torch.equal(p, q)
    assert Adam([{"params": [torch.nn.Parameter(torch.zeros(2))], "lr": 0.1}]).param_groups[0]["lr"] == 0.1


def test_sliced_spmm_plan_is_a_partition_of_the_matrix(monkeypatch):
    """CsrPlan.sliced (the host side of mi_spmm_sliced): narrow tasks (a lane group owns two rows), wide tasks (rows of more
    than WIDE_MIN nonzeros, all groups stride one range), hubs apart; every non-hub edge exactly once, inside an owner's
    range slice by slice; A (and A^T) reassembled from the plan exactly as the kernel reads it is A."""
    import numpy as np

    from recsys_benchmark_amd import _kernels as K

    monkeypatch.setattr(K, "HUB_DEGREE", 60)
    monkeypatch.setattr(K.CsrPlan, "WIDE_MIN", 24)
    monkeypatch.setattr(K.CsrPlan, "SLICE_BYTES", 64 * 4 * 16)         # 16 rows of X (D = 64) per slice
    gen = torch.Generator().manual_seed(0)
    n, m, nnz = 300, 260, 6000
    r = torch.randint(0, n, (nnz,), generator=gen)
    c = (m * torch.rand(nnz, generator=gen).pow(2)).long().clamp_(max=m - 1)
    A = torch.sparse_coo_tensor(torch.stack([r, c]), torch.randn(nnz, generator=gen), (n, m)).coalesce().to_sparse_csr()
    plan = K.CsrPlan(A.crow_indices(), A.col_indices(), tuple(A.shape))
    for D in (64, 16):
        for tr in (False, True):
            sp = plan.sliced(D, tr)
            NPW = 256 // D
            val = plan.transposed_values(A.values()) if tr else A.values()
            ev = plan.sliced_values(val, D, tr)
            T = sp["T"]
            nr, nc = (m, n) if tr else (n, m)
            tptr, trows, twide = sp["tptr"].numpy(), sp["trows"].numpy(), sp["twide"].numpy()
            ecol = sp["ecol"].numpy().astype(np.int64) & 0xFFFFFFFF
            deg = np.diff((plan.crow_t if tr else plan.crow).numpy())
            hubs = sp["long_rows"].numpy()
            rows_seen = trows[trows >= 0]
            assert sorted(rows_seen.tolist() + hubs.tolist()) == list(range(nr)), "tasks + hubs cover every row once"
            assert (deg[hubs] > 60).all() and (deg[rows_seen] <= 60).all()
            assert twide.sum() > 0 and (twide == 0).sum() > 0
            dense = torch.zeros(nr, nc)
            sr = sp["slice_rows"]
            for t in range(T):
                groups = [0] if twide[t] else range(NPW)
                for k in groups:
                    lo, hi = (tptr[t, 0], tptr[t, 1]) if twide[t] else (tptr[t, k], tptr[t, k + 1])
                    own = [trows[t, j * NPW + k] for j in range(4)]
                    if twide[t]:
                        assert all(r < 0 or deg[r] > 24 for r in own) and own[0] >= 0 and (trows[t, 1:NPW] < 0).all()
                    last_slice = -1
                    for e in range(lo, hi):
                        cc, j = int(ecol[e]) & ((1 << 28) - 1), int(ecol[e]) >> 28
                        assert cc // sr >= last_slice, "an owner's edges ascend slice by slice"
                        last_slice = cc // sr
                        row = own[j]
                        assert row >= 0
                        dense[row, cc] += ev[e]
                if twide[t]:
                    assert (tptr[t, 1:] == tptr[t, 1]).all()
            crow, col = (plan.crow_t if tr else plan.crow).numpy(), (plan.col_t if tr else plan.col).numpy()
            for hr in hubs.tolist():
                for e in range(crow[hr], crow[hr + 1]):
                    dense[hr, col[e]] += val[e]
            assert torch.allclose(dense, A.to_dense().t() if tr else A.to_dense())


def test_csr_plan_caches_never_serve_a_recycled_address():
    """CsrPlan.transposed_values / csr_plan are keyed on tensor addresses: consecutive SparseDropout draws (each a new
    tensor, typically at the address the previous one freed) must each get their own permuted values, and a second
    graph of the same shape allocated where a dead one lived must get its own plan."""
    from recsys_benchmark_amd import _kernels as K
    from recsys_benchmark_amd.layers import SparseDropout

    torch.manual_seed(0)
    n = 120

    def graph(seed):
        g = torch.Generator().manual_seed(seed)
        a = (torch.rand(n, n, generator=g) < 0.06).float()
        a = ((a + a.t()) > 0).float() * torch.rand(n, n, generator=g)
        return a.to_sparse_csr()

    csr = graph(1)
    plan = K.csr_plan(csr)
    drop = SparseDropout(0.5).train()
    for _ in range(10):
        d = drop(csr)
        v = d.values()
        assert torch.equal(plan.transposed_values(v), v.index_select(0, plan.perm))
        del d, v
    # same tensor again -> served from the cache (identity), in-place edit -> recomputed
    v = csr.values().clone()
    t1 = plan.transposed_values(v)
    assert plan.transposed_values(v) is t1
    v.mul_(2.0)
    assert torch.equal(plan.transposed_values(v), v.index_select(0, plan.perm))
    # plans keep their source index tensors alive, so a new graph cannot alias a cached key
    K._plans.clear()
    for seed in range(2, 8):
        g = graph(seed)
        p = K.csr_plan(g)
        assert torch.equal(p.crow.long(), g.crow_indices()) and torch.equal(p.col.long(), g.col_indices())
        del g, p


def test_bucket_capacity_accounts_for_fields_with_fewer_values_than_ranks():
    """A Criteo field with 3 values sends its whole column to 3 owners: the per-field capacity must cover the most loaded
    owner's EXPECTED load with headroom, which the plain mean n / world does not (advisor finding, round 1)."""
    from recsys_benchmark_amd.sharded import bucket_capacity, expected_peak_load, field_bucket_capacity

    criteo = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
              27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
    B, world = 4096, 8
    gen = torch.Generator().manual_seed(0)
    off = torch.tensor([0] + criteo[:-1]).cumsum(0)
    peak = expected_peak_load(criteo, B, world)
    assert peak > B * len(criteo) / world                      # the low-cardinality fields concentrate
    worst = 0
    for _ in range(20):
        rows = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in criteo], 1) + off
        worst = max(worst, int(torch.bincount((rows % world).view(-1), minlength=world).max()))
    cap = field_bucket_capacity(criteo, B, world, 1.25)
    assert worst <= cap, (worst, cap)
    assert abs(worst - peak) < 6 * peak ** 0.5 + 64            # and the estimate is the right size
    # adversarial: every other field has ONE value and they all land on owner 0 (offsets 0, 8, 16, ...) — the old sizing
    # would overflow, this one must not
    tiny = [1, 7] * 8
    rows_per_owner = expected_peak_load(tiny, B, world)
    assert rows_per_owner > bucket_capacity(B * len(tiny), world, 1.25)
    assert field_bucket_capacity(tiny, B, world, 1.25) >= rows_per_owner
    assert field_bucket_capacity(tiny, B, 1, 1.25) == B * len(tiny)


def test_dropout_bit_generator_restatement_rate_and_layout():
    """tests/tail_helpers.py restates the fused tail's keep-bit generator (csrc/tail.hip k_tail_dropmask): deterministic in
    (seed, salt), the right keep rate, one decision per element, salt and seed both matter."""
    from tail_helpers import tail_keep_scale

    a = tail_keep_scale(123, 7919, 512, 400, 0.5)
    assert a.shape == (512, 400) and torch.equal(a, tail_keep_scale(123, 7919, 512, 400, 0.5))
    assert set(a.unique().tolist()) == {0.0, 2.0}
    assert abs(float((a > 0).float().mean()) - 0.5) < 0.01
    assert not torch.equal(a, tail_keep_scale(124, 7919, 512, 400, 0.5))
    assert not torch.equal(a, tail_keep_scale(123, 2 * 7919, 512, 400, 0.5))
    b = tail_keep_scale(5, 1, 256, 64, 0.1)
    assert abs(float((b > 0).float().mean()) - 0.9) < 0.01 and abs(float(b.max()) - 1.0 / 0.9) < 1e-6
    assert torch.equal(tail_keep_scale(5, 1, 256, 64, 0.0), torch.ones(256, 64))


def test_bench_cpu_baseline_protocol_on_a_small_shape():
    """bench.py's CPU legs: 3 warm-up + up to 10 timed iterations inside a time budget, physical-core thread count, the
    like-for-like line first and the reference-default lines after it."""
    import bench

    phys, logical = bench.physical_cores()
    assert 1 <= phys <= logical
    old = torch.get_num_threads()
    try:
        r = bench.cpu_baseline([50, 30, 2000, 10000], 16, [32, 32], 256, 0.5, budget_s=4.0)
    finally:
        torch.set_num_threads(old)
    assert r["kind"] == "port" and r["cores"] == phys and r["unit"] == "samples/s" and r["value"] > 0
    assert len(r["lines"]) == 4 and "row-form" in r["lines"][0]["what"] and "Adam" in r["lines"][2]["what"]
    assert all(2 <= ln["timed_iters"] <= 10 for ln in r["lines"])
    med, n = bench.timed_cpu(lambda: None, warm=1, timed=5, budget_s=1.0)
    assert n == 5 and med >= 0


def test_dhe_per_item_hash_table_is_the_references():
    """Host logic of use_universal_hash=False (dh_embedding.py:155-196): the [N, k] feature table built at construction is
    bit-identical to the reference's, and the constructor leaves torch's global generator where the reference leaves it."""
    g = load_golden("dhe_peritem")
    c0 = DHEmbedding.COUNTER
    try:
        DHEmbedding.COUNTER = int(g["prefix"])
        torch.manual_seed(2023)
        emb = DHEmbedding(g["field_dims"].tolist(), int(g["D"]), None, int(g["k"]), g["hidden"].tolist(), cached=True,
                          use_universal_hash=False)
        after = torch.randint(0, 2**31 - 1, (4,))
    finally:
        DHEmbedding.COUNTER = c0
    assert torch.equal(emb._cache, g.t("cache"))
    assert torch.equal(after, g.t("after_init"))
    assert emb._cache.min() >= -1 and emb._cache.max() <= 1
    assert sorted(emb.state_dict().keys()) == sorted(["_extra_state"] + ["_seq." + k for k in g.group("param/_seq.")])


@pytest.mark.parametrize("name", ["ttinit_approx_uniform_r4x6", "ttinit_approx_uniform_r2x3"])
def test_tt_approx_uniform_cores_are_the_references(name):
    """Host logic of weight_dist='approx-uniform' (tt_embedding_ops.py:861-986): same numpy / `random` draw order, so the
    seeded cores are bit-identical; two cores is refused like the reference's assert."""
    import random

    import numpy as np

    from recsys_benchmark_amd.embeddings import TTRecTorch

    g = load_golden(name)
    ps, qs, rs = g["tt_p_shapes"].tolist(), g["tt_q_shapes"].tolist(), g["tt_ranks"].tolist()
    np.random.seed(2023), torch.manual_seed(2023), random.seed(2023)
    explicit = name.endswith("r2x3")
    emb = TTRecTorch(int(g["num_item"]), int(g["hidden"]), rs[1:-1], tt_p_shapes=ps if explicit else None,
                     tt_q_shapes=qs if explicit else None, weight_dist="approx-uniform")
    assert emb.tt_p_shapes == ps and emb.tt_q_shapes == qs
    for i, c in enumerate(emb.tt_cores):
        assert c.dtype == torch.float32 and torch.equal(c.data, g.t(f"param/tt_cores.{i}")), f"core {i}"
    with pytest.raises(AssertionError):
        TTRecTorch(100, 8, [4], tt_p_shapes=[10, 10], tt_q_shapes=[2, 4], weight_dist="approx-uniform")


def test_multi_problem_launch_is_cut_for_the_whole_launch(monkeypatch):
    """Host logic of mi_gemm_f32_multi's K-slices (csrc/gemm.hip auto_splitk through mi_gemm_f32_multi_plan: arithmetic
    only, runs without a GPU): one slice count for the launch, the one that loads 256 CUs most evenly with the atomics of
    every extra slice priced in — the C2 tail's three gradients (147 tiles of 64 x 64, 128 k-tiles) get 5 slices = 735
    workgroups = 2.9 per CU (round 2's round(640 / tiles) = 4 put 3 on most CUs and 2 on the rest); a slice keeps >= 4
    k-tiles; explicit values are kept; deterministic mode does not split; the weight-gradient form goes to the LDS-DMA
    kernel exactly when every problem of the launch fits it."""
    from recsys_benchmark_amd import _kernels

    def q(M, N, K, **kw):
        return dict(A=torch.zeros(4), B=torch.zeros(4), C=torch.zeros(4), M=M, N=N, K=K, lda=M, ldb=N, ldc=N, **kw)

    B, d, E, r = 4096, 352, 4, 64
    tail = [q(400, 416, B), q(400, 400, B), q(400, 400, B)]
    assert _kernels.gemm_multi_plan(tail, transA=True) == (1, 735, [5, 5, 5])
    assert _kernels.gemm_multi_plan(tail, transA=False) == (0, 735, [5, 5, 5])
    layer = [q(E * r, d, B), q(r, r, B, batch=E, sA=r, sB=r, sC=r * r), q(d, r, B, batch=E, sB=r, sC=d * r), q(E, d, B)]
    kind, wgs, sk = _kernels.gemm_multi_plan(layer * 3, transA=True)          # the 12 weight gradients of a DCN-Mix backward
    assert kind == 1 and len(set(sk)) == 1 and 500 <= wgs <= 800, (kind, wgs, sk)            # 174 tiles
    assert _kernels.gemm_multi_plan([q(64, 64, 96)], transA=True) == (1, 1, [1])              # 3 k-tiles: never below 4 per slice
    assert _kernels.gemm_multi_plan([q(64, 64, B, splitk=7)], transA=True) == (1, 7, [7])
    assert _kernels.gemm_multi_plan([q(64, 64, B), q(0, 64, B)], transA=True)[2][1] == 0     # an empty problem
    # a K tail or a ragged width in ANY problem keeps the whole launch on the general kernel
    assert _kernels.gemm_multi_plan([q(100, 70, 45)], transA=True)[0] == 0
    assert _kernels.gemm_multi_plan(tail + [q(102, 64, B)], transA=True)[0] == 0
    monkeypatch.setattr(_kernels, "DETERMINISTIC", True)
    assert _kernels.gemm_multi_plan(layer * 3, transA=True)[2] == [1] * 12
    assert _kernels.gemm_multi_plan(tail, transA=False)[2] == [1] * 3


def test_deterministic_switch_restores_the_fused_tail_setting_and_refuses_crossnet_training():
    """use_deterministic_algorithms(False) must give back the mlp.FUSED_TAIL that was in effect before (round 2 switched
    the default-on fused tail OFF for every later run), and DCN training under the switch must refuse: its backward
    accumulates with float atomics, which the switch promises not to use."""
    import pytest

    from recsys_benchmark_amd import _kernels, mlp
    from recsys_benchmark_amd.dcn import DCN_Mix

    before = mlp.FUSED_TAIL
    try:
        for setting in (True, False):
            mlp.FUSED_TAIL = setting
            pkg.use_deterministic_algorithms(True)
            assert mlp.FUSED_TAIL is True and _kernels.DETERMINISTIC
            pkg.use_deterministic_algorithms(True)          # twice: the remembered setting must survive
            pkg.use_deterministic_algorithms(False)
            assert mlp.FUSED_TAIL is setting and not _kernels.DETERMINISTIC
        pkg.use_deterministic_algorithms(False)             # off while off: nothing to restore, nothing changes
        assert mlp.FUSED_TAIL is False
        m = DCN_Mix([3, 4], 4, [8], num_layers=1, num_experts=2, rank=2)
        pkg.use_deterministic_algorithms(True)
        with pytest.raises(NotImplementedError):
            m(torch.tensor([[0, 1]]))
    finally:
        pkg.use_deterministic_algorithms(False)
        mlp.FUSED_TAIL = before
