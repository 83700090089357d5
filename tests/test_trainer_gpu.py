"""GPU: the CTR training loop around the path (reference: src/trainer/deepfm.py:17-139).  A hipGraph-replayed training step
must leave the model and both optimizers exactly where the same steps launched eagerly leave them (same kernels, same
order; the only run-to-run freedom is the order of float atomics in the dense first-order-table gradient)."""
import copy

import pytest
import torch

from conftest import assert_close

import recsys_benchmark_amd as pkg
from recsys_benchmark_amd import trainer
from recsys_benchmark_amd.optim import get_optimizers

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[True, False], ids=["own-tail", "library-tail"])
def _both_tails(request, monkeypatch):
    """Every test of this file runs with the MLP tail on the own fused kernels (the default) and on the general path
    (library products + the fused BatchNorm passes), so both stay covered whatever the default is."""
    from recsys_benchmark_amd import mlp as _mlp_mod

    monkeypatch.setattr(_mlp_mod, "FUSED_TAIL", request.param)
DEV = torch.device("cuda", 0)
DIMS = [50, 3, 1000, 7, 200]


def _batches(n, B, seed):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in DIMS], 1)
        out.append((x, (torch.rand(B, generator=gen) < 0.3).float()))
    return out


def _model(sparse, fc_sparse=False):
    torch.manual_seed(3)
    return pkg.DeepFM(DIMS, 16, [64, 32], p_dropout=0.0, use_batchnorm=True,
                      embedding_config={"name": "vanilla", "sparse": sparse}, fc_sparse=fc_sparse).to(DEV)


@pytest.mark.parametrize("sparse,fc_sparse", [(True, False), (True, True), (False, False)])
def test_graphed_step_equals_eager_steps(sparse, fc_sparse):
    cfg = {"sparse": sparse, "optimizer": "adam", "learning_rate": 1e-2, "weight_decay": 1e-6}
    ref_model = _model(sparse, fc_sparse)
    model = copy.deepcopy(ref_model)
    ref_step = trainer.GraphedTrainStep(ref_model, get_optimizers(ref_model, cfg), use_graph=False)
    step = trainer.GraphedTrainStep(model, get_optimizers(model, cfg), warmup=2)
    data = _batches(9, 256, 11)
    for x, y in data:
        ref_step(x.to(DEV), y.to(DEV))
        step(x.to(DEV), y.to(DEV))
    assert step._graph is not None, "the step was never captured"
    assert_close(step.loss_sum, ref_step.loss_sum, 1e-5, 1e-5, "accumulated loss")
    # Adam divides by sqrt(v): where a gradient is almost zero, the last-bit freedom of the float atomics (dense table
    # gradients, column sums of the MLP backward) comes out amplified in a handful of elements — 1e-4 absolute is
    # 1 % of one lr-sized step; a stale buffer or a wrong step count would be off by whole steps (1e-2)
    for (k, a), (_, b) in zip(model.state_dict().items(), ref_model.state_dict().items()):
        assert_close(a, b, 5e-3, 1e-4, k)
    for oa, ob in zip(step.optimizers, ref_step.optimizers):
        for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
            if not oa.state[pa] and not ob.state[pb]:
                continue                                   # never received a gradient (DeepFM.linear_layer is unused)
            for key in ("step", "exp_avg", "exp_avg_sq"):
                assert_close(torch.as_tensor(oa.state[pa][key]).float(), torch.as_tensor(ob.state[pb][key]).float(), 1e-3, 1e-7,
                             key)
    # a batch of another shape runs eagerly and the graph stays valid for the next full batch
    x, y = _batches(1, 100, 5)[0]
    for s in (ref_step, step):
        s(x.to(DEV), y.to(DEV))
        s(data[0][0].to(DEV), data[0][1].to(DEV))
    assert_close(step.loss_sum, ref_step.loss_sum, 1e-5, 1e-5, "accumulated loss after a ragged batch")
    assert_close(model.embedding.get_weight(), ref_model.embedding.get_weight(), 5e-3, 1e-4, "table after a ragged batch")


def test_train_epoch_matches_a_hand_written_loop_and_learns():
    cfg = {"sparse": True, "optimizer": "adam", "learning_rate": 1e-2, "weight_decay": 1e-6}
    model = _model(True)
    ref = copy.deepcopy(model)
    data = _batches(6, 128, 2) + _batches(1, 40, 3)          # ragged last batch
    opts = get_optimizers(model, cfg)
    step = trainer.GraphedTrainStep(model, opts)
    first = trainer.train_epoch(data, model, opts, device=DEV, log_step=3, step=step)
    # the reference's loop, spelled out with stock criterion on the same kernels
    ropts = get_optimizers(ref, cfg)
    crit = torch.nn.BCEWithLogitsLoss()
    total = 0.0
    ref.train()
    for x, y in data:
        out = ref(x.to(DEV))
        loss = crit(out, y.to(DEV).float())
        for o in ropts:
            o.zero_grad()
        loss.backward()
        for o in ropts:
            o.step()
        total += loss.item()
    # seven Adam steps apart the two runs differ by the float-atomics noise Adam amplifies (see the test above): 2e-4 on a
    # mean loss of ~0.7; a skipped, repeated or stale batch would move it by 1e-2
    assert abs(first["loss"] - total / len(data)) < 2e-4
    later = [trainer.train_epoch(data, model, opts, device=DEV, log_step=0, step=step)["loss"] for _ in range(15)][-1]
    assert later < first["loss"] * 0.8, (first, later)


def test_validate_epoch_matches_sklearn():
    from sklearn.metrics import log_loss, roc_auc_score

    model = _model(False)
    data = _batches(5, 200, 8) + _batches(1, 33, 9)
    res = trainer.validate_epoch(data, model, device=DEV)
    model.eval()
    with torch.no_grad():
        pred = torch.cat([torch.sigmoid(model(x.to(DEV))).cpu() for x, _ in data]).double()
    true = torch.cat([y for _, y in data])
    assert abs(res["auc"] - roc_auc_score(true.tolist(), pred.tolist())) < 1e-9
    assert abs(res["log_loss"] - log_loss(true.numpy(), pred.numpy())) < 1e-5


# ------------------------------------------------------------------ LightGCN loops (src/trainer/lightgcn.py)
class _ToyCF:
    """The two methods of the reference's CFGraphDataset the trainer uses."""

    def __init__(self, num_user=60, num_item=90, seed=0):
        from recsys_benchmark_amd.graph_utils import calculate_sparse_graph_adj_norm

        gen = torch.Generator().manual_seed(seed)
        self.num_user, self.num_item = num_user, num_item
        self.graph = {u: sorted(set(torch.randint(0, num_item, (int(torch.randint(2, 12, (1,), generator=gen)),),
                                                  generator=gen).tolist())) for u in range(num_user)}
        self.adj = calculate_sparse_graph_adj_norm(self.graph, num_item, num_user)

    def get_norm_adj(self):
        return self.adj

    def get_graph(self):
        return self.graph

    def triples(self, n, B, seed):
        gen = torch.Generator().manual_seed(seed)
        return [(torch.randint(0, self.num_user, (B,), generator=gen), torch.randint(0, self.num_item, (B,), generator=gen),
                 torch.randint(0, self.num_item, (B,), generator=gen)) for _ in range(n)]


class _Loader(list):
    dataset = None


@pytest.mark.parametrize("info_nce_weight", [0.0, 0.1])
def test_cf_train_epoch_graph_equals_eager_and_reference_ops(info_nce_weight):
    from oracle import reference_ops as ro
    from recsys_benchmark_amd.optim import Adam

    ds = _ToyCF()
    torch.manual_seed(1)
    model = pkg.LightGCN(ds.num_user, ds.num_item, num_layers=2, hidden_size=16).to(DEV)
    eager, stock = copy.deepcopy(model), copy.deepcopy(model)
    data = _Loader(ds.triples(7, 64, 3) + ds.triples(1, 20, 4))
    data.dataset = ds
    wd = 1e-3
    got = trainer.train_epoch_cf(data, model, Adam(model.parameters(), lr=1e-2), device=DEV, log_step=4, weight_decay=wd,
                                 info_nce_weight=info_nce_weight)
    estep = trainer.GraphedCFTrainStep(eager, ds.adj.to(DEV), Adam(eager.parameters(), lr=1e-2), wd, info_nce_weight,
                                       use_graph=False)
    want = trainer.train_epoch_cf(data, eager, None, device=DEV, log_step=0, step=estep)
    # the reference's _train_step in stock torch ops (oracle restatements of the losses) on the same batches
    opt = torch.optim.Adam(stock.parameters(), lr=1e-2)
    adj = ds.adj.to(DEV)
    tot = torch.zeros(4)
    for users, pos, neg in data:
        users, pos, neg = users.to(DEV), pos.to(DEV), neg.to(DEV)
        au, ai = stock(adj)
        rec = ro.bpr_loss(au[users], ai[pos], ai[neg])
        reg = stock.get_reg_loss(users, pos, neg)
        cl = torch.zeros((), device=DEV)
        if info_nce_weight:
            view = torch.cat([au[torch.unique(users)], ai[torch.unique(pos)]])
            cl = ro.info_nce(view, view, 0.2) * info_nce_weight
        loss = rec + wd * reg + cl
        opt.zero_grad()
        loss.backward()
        opt.step()
        tot += torch.stack([loss, rec, reg, cl]).detach().cpu()
    for key, ref in zip(("loss", "rec_loss", "reg_loss", "cl_loss"), (tot / len(data)).tolist()):
        # fp32 losses over 8 Adam steps; the dense table gradient is summed with float atomics
        assert abs(got[key] - want[key]) < 1e-4 * max(1.0, abs(want[key])), (key, got[key], want[key])
        assert abs(got[key] - ref) < 1e-4 * max(1.0, abs(ref)), (key, got[key], ref)
    for (k, a), (_, b), (_, c) in zip(model.state_dict().items(), eager.state_dict().items(), stock.state_dict().items()):
        assert_close(a, b, 5e-3, 1e-4, k + " graph vs eager")
        assert_close(a, c, 5e-3, 1e-4, k + " vs stock ops")


def test_cf_validate_epoch_matches_reference_procedure():
    from oracle import reference_ops as ro

    ds = _ToyCF(seed=5)
    torch.manual_seed(2)
    model = pkg.LightGCN(ds.num_user, ds.num_item, num_layers=3, hidden_size=16).to(DEV)
    gen = torch.Generator().manual_seed(6)
    val = [(torch.arange(s, min(s + 25, ds.num_user)),
            [set(torch.randint(0, ds.num_item, (int(torch.randint(1, 9, (1,), generator=gen)),), generator=gen).tolist())
             for _ in range(s, min(s + 25, ds.num_user))]) for s in range(0, ds.num_user, 25)]
    k = 10
    got = trainer.validate_epoch_cf(ds, val, model, device=DEV, k=k, metrics=["ndcg", "recall"])
    model.eval()
    with torch.no_grad():
        ue, ie = model(ds.adj.to(DEV))
        preds, truths = [], []
        for users, pos in val:
            scores = (ue[users.to(DEV)] @ ie.T).cpu()
            for row, u in enumerate(users.tolist()):
                scores[row, ds.graph[u]] = float("-inf")            # src/trainer/lightgcn.py:126-133
            preds.extend(torch.topk(scores, k)[1].tolist())
            truths.extend(pos)
    ndcg, recall = ro.ndcg_recall(preds, truths, k)
    assert abs(got["ndcg"] - ndcg) < 1e-9 and abs(got["recall"] - recall) < 1e-9
    assert set(trainer.validate_epoch_cf(ds, val, model, device=DEV, k=k)) == {"ndcg"}


def test_train_epoch_cerp_adds_the_prune_loss_and_reports_sparsity():
    """src/trainer/deepfm.py:142-248 on a CERP table: one-graph steps equal eager steps, the loss is log-loss +
    weight * prune loss, and the early return fires once the target sparsity is reached."""
    torch.manual_seed(5)
    cfg = {"name": "cerp", "bucket_size": 40}
    model = pkg.DeepFM(DIMS, 16, [32], p_dropout=0.0, use_batchnorm=True, embedding_config=cfg).to(DEV)
    eager = copy.deepcopy(model)
    data = _batches(8, 128, 21)
    w = 1e-4
    from recsys_benchmark_amd.optim import Adam

    gstep = trainer.GraphedTrainStep(model, Adam(model.parameters(), lr=1e-2), extra_loss=lambda: model.embedding.get_prune_loss(),
                                     extra_weight=w)
    got = trainer.train_epoch_cerp(data, model, None, device=DEV, log_step=0, prune_loss_weight=w, step=gstep)
    assert gstep._graph is not None, "the CERP step was never captured"
    estep = trainer.GraphedTrainStep(eager, Adam(eager.parameters(), lr=1e-2), use_graph=False,
                                     extra_loss=lambda: eager.embedding.get_prune_loss(), extra_weight=w)
    want = trainer.train_epoch_cerp(data, eager, None, device=DEV, log_step=0, prune_loss_weight=w, step=estep)
    assert set(got) == {"loss", "prune_loss", "log_loss", "sparsity", "num_params"}
    for key in ("loss", "prune_loss", "log_loss"):
        assert abs(got[key] - want[key]) < 2e-4 * max(1.0, abs(want[key])), (key, got[key], want[key])
    assert abs(got["loss"] - (got["log_loss"] + w * got["prune_loss"])) < 1e-5 and got["prune_loss"] < 0
    assert got["num_params"] == want["num_params"] and 0.0 <= got["sparsity"] < 1.0
    # target sparsity already met: returns at the first logging step with running sums
    # (a stock torch optimizer keeps its step count on the host: such a step is never captured, it runs eagerly)
    with pytest.warns(UserWarning, match="capturable=False"):
        early = trainer.train_epoch_cerp(data, model, torch.optim.Adam(model.parameters(), lr=1e-2), device=DEV, log_step=1,
                                         prune_loss_weight=w, target_sparsity=-1.0)
    assert early["sparsity"] >= -1.0 and early["log_loss"] > 0


@pytest.mark.parametrize("emb_cfg", [
    {"name": "qr", "divider": 5, "operation": "mult"}, {"name": "qr", "divider": 3, "operation": "cat"},
    {"name": "cerp", "bucket_size": 40}, {"name": "pep", "threshold_type": "feature_dim"}, {"name": "qat", "n_bits": 8},
    {"name": "tt_emb_torch", "tt_ranks": [4, 4]}, {"name": "dhe", "inp_size": 64, "hidden_sizes": [32]},
    {"name": "deepfm_optembed", "t_init": 0.0},
], ids=lambda c: c["name"] + str(c.get("operation", "")))
def test_every_table_kind_trains_inside_one_graph(emb_cfg, tmp_path):
    """The step with each compressed table is captured (no host sync, no data-dependent shape anywhere on the path) and
    replays to the same losses as eager launches of the same kernels."""
    from recsys_benchmark_amd.optim import Adam

    cfg = dict(emb_cfg)
    if cfg["name"] == "pep":
        cfg["checkpoint_weight_dir"] = str(tmp_path)
    torch.manual_seed(9)
    model = pkg.DeepFM(DIMS, 16, [32], p_dropout=0.0, use_batchnorm=True, embedding_config=cfg).to(DEV)
    eager = copy.deepcopy(model)
    gstep = trainer.GraphedTrainStep(model, Adam(model.parameters(), lr=1e-2))
    estep = trainer.GraphedTrainStep(eager, Adam(eager.parameters(), lr=1e-2), use_graph=False)
    draws = cfg["name"] in ("qat", "deepfm_optembed")          # stochastic rounding / sampled masks: own random streams
    for x, y in _batches(6, 128, 31):
        gstep(x.to(DEV), y.to(DEV))
        estep(x.to(DEV), y.to(DEV))
    assert gstep._graph is not None, "never captured"
    assert torch.isfinite(gstep.loss_sum)
    if not draws:
        assert_close(gstep.loss_sum, estep.loss_sum, 2e-4, 1e-5, "accumulated loss, graph vs eager")


def test_end_to_end_training_learns_a_planted_signal():
    """train_epoch + validate_epoch on synthetic clicks drawn from a planted first-order model: the validation AUC of the
    trained DeepFM (row-form table gradients, SparseAdam + Adam, one-graph steps) climbs well above chance."""
    gen = torch.Generator().manual_seed(13)
    dims = [30, 8, 200, 5, 60]
    planted = [torch.randn(d, generator=gen) for d in dims]

    def draw(n, B):
        out = []
        for _ in range(n):
            x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1)
            logit = sum(w[x[:, f]] for f, w in enumerate(planted)) * 0.8
            out.append((x, (torch.rand(B, generator=gen) < torch.sigmoid(logit)).float()))
        return out

    train, val = draw(40, 256), draw(8, 256)
    torch.manual_seed(1)
    model = pkg.DeepFM(dims, 8, [32, 16], p_dropout=0.1, use_batchnorm=True,
                       embedding_config={"name": "vanilla", "sparse": True}, fc_sparse=True).to(DEV)
    opts = get_optimizers(model, {"sparse": True, "optimizer": "adam", "learning_rate": 5e-3, "weight_decay": 1e-6})
    step = trainer.GraphedTrainStep(model, opts)
    before = trainer.validate_epoch(val, model, device=DEV)
    losses_seen = [trainer.train_epoch(train, model, opts, device=DEV, log_step=0, step=step)["loss"] for _ in range(6)]
    after = trainer.validate_epoch(val, model, device=DEV)
    assert step._graph is not None
    assert losses_seen[-1] < losses_seen[0] - 0.05, losses_seen
    assert after["auc"] > 0.75 > before["auc"] - 0.2 and after["log_loss"] < before["log_loss"], (before, after)
    pkg.check_index_errors()


def test_train_epoch_pep_reports_sparsity_and_stops_at_the_target(tmp_path):
    from recsys_benchmark_amd.optim import Adam

    ds = _ToyCF(seed=3)
    torch.manual_seed(4)
    cfg = {"name": "pep", "threshold_type": "feature_dim", "checkpoint_weight_dir": str(tmp_path)}
    model = pkg.LightGCN(ds.num_user, ds.num_item, num_layers=2, hidden_size=16, embedding_config=cfg).to(DEV)
    data = _Loader(ds.triples(6, 64, 8))
    data.dataset = ds
    out = trainer.train_epoch_pep(data, model, Adam(model.parameters(), lr=1e-2), device=DEV, log_step=2, weight_decay=1e-3,
                                  target_sparsity=2.0)
    assert set(out) == {"loss", "rec_loss", "reg_loss", "cl_loss", "sparsity", "num_params"}
    assert 0.0 <= out["sparsity"] <= 1.0 and out["num_params"] > 0 and out["rec_loss"] > 0
    # a target already met: the epoch ends at the first logging step (one batch)
    step = trainer.GraphedCFTrainStep(model, ds.adj.to(DEV), Adam(model.parameters(), lr=1e-2), 1e-3)
    trainer.train_epoch_pep(data, model, None, device=DEV, log_step=1, target_sparsity=-1.0, step=step)
    assert step.steps == 1


@pytest.mark.parametrize("cls_name", ["DCN_Mix", "DCNv2"])
def test_dcn_with_row_form_gradients_trains_like_torch_optimizers(cls_name):
    """The DCN models note the field layout of their ids too: SparseAdam sorts them field by field, and the parameters
    follow the same model stepped by torch.optim.SparseAdam + torch.optim.Adam."""
    from recsys_benchmark_amd import _kernels, dcn

    torch.manual_seed(6)
    make = getattr(dcn, cls_name)
    kw = dict(num_layers=2, embedding_config={"name": "vanilla", "sparse": True}, p_dropout=0.0)
    if cls_name == "DCN_Mix":
        kw.update(num_experts=2, rank=8)
    model = make(DIMS, 8, [32], **kw).to(DEV)
    twin = copy.deepcopy(model)
    cfg = {"sparse": True, "optimizer": "adam", "learning_rate": 1e-2, "weight_decay": 1e-6}
    mine = trainer.GraphedTrainStep(model, get_optimizers(model, cfg))
    dense = [p for n, p in twin.named_parameters() if "embedding." not in n]
    theirs = [torch.optim.SparseAdam(list(twin.embedding.parameters()), lr=1e-2),
              torch.optim.Adam(dense, lr=1e-2, weight_decay=1e-6)]
    crit = torch.nn.BCEWithLogitsLoss()
    hits = 0
    for x, y in _batches(5, 128, 41):
        x, y = x.to(DEV), y.to(DEV)
        mine(x, y)
        for o in theirs:
            o.zero_grad()
        crit(twin(x), y).backward()
        rows = twin.embedding.get_weight().grad._indices()[0]
        hits += _kernels.sort_field_rows(rows, sum(DIMS)) is not None
        for o in theirs:
            o.step()
    assert hits == 5, "the ids' field layout was not found for the optimizer's sort"
    # five Adam steps (lr 1e-2) apart; the cross network's weight gradients are split-K sums with float atomics, whose
    # last-bit freedom Adam amplifies where a gradient is almost zero (seen: 5e-5 on 2 of 10 080 elements in one run of
    # ten) — 2e-4 is 2 % of one step, a wrong sort or a stale buffer would be off by whole steps
    for (k, a), (_, b) in zip(model.state_dict().items(), twin.state_dict().items()):
        assert_close(a, b, 5e-3, 2e-4, k)


def test_adam_amplified_differences_are_noise_around_the_float64_run():
    """The tolerances of the training-equivalence tests above (5e-3 relative / 1 % of one lr-sized step absolute) rest on
    one claim: float32 summation order (float atomics, split-K) leaves last-bit noise in near-zero gradients, which Adam's
    1/sqrt(v) normalisation amplifies — noise AROUND the exact trajectory, not a drift away from it.  Tested as such: the
    same nine steps in FLOAT64 through the oracle's op sequence (oracle/reference_ops.py deepfm_forward = the reference's
    src/models/deepfm.py:79-105) and torch.optim.Adam(weight_decay) — the reference's dense configuration,
    src/models/deepfm.py:186-193 — and BOTH float32 runs (hipGraph replay and eager) must sit within that tolerance of the
    float64 parameters."""
    from oracle import reference_ops as ro

    cfg = {"sparse": False, "optimizer": "adam", "learning_rate": 1e-2, "weight_decay": 1e-6}
    eager_model = _model(False)
    graph_model = copy.deepcopy(eager_model)
    p64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.detach().clone()) for k, v in eager_model.state_dict().items()}
    leaves = []
    for k, v in p64.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
            leaves.append(v)
    opt64 = torch.optim.Adam(leaves, lr=1e-2, weight_decay=1e-6)
    eager = trainer.GraphedTrainStep(eager_model, get_optimizers(eager_model, cfg), use_graph=False)
    graphed = trainer.GraphedTrainStep(graph_model, get_optimizers(graph_model, cfg), warmup=2)
    crit = torch.nn.BCEWithLogitsLoss()
    for x, y in _batches(9, 256, 11):
        x, y = x.to(DEV), y.to(DEV)
        eager(x, y)
        graphed(x, y)
        opt64.zero_grad()
        crit(ro.deepfm_forward(x, p64, 2, True, True), y.double()).backward()
        opt64.step()
    assert graphed._graph is not None
    sd_e, sd_g = eager_model.state_dict(), graph_model.state_dict()
    for k, v in p64.items():
        if not (v.is_floating_point() and v.requires_grad) or k == "linear_layer.weight":
            continue
        for name, sd in (("eager", sd_e), ("graph", sd_g)):
            assert_close(sd[k].double(), v.detach(), 5e-3, 1e-4, f"{k} ({name} float32 run vs the float64 run)")
