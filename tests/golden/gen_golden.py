#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference, CPU PyTorch):

    python tests/golden/gen_golden.py

The reference's modules are imported unmodified; `loguru` (not installed here) is
replaced by a no-op stand-in placed in sys.modules before the import (SURVEY.md
§8c).  Only arrays leave this script: inputs, parameters (state_dict tensors),
outputs and parameter gradients, saved with numpy — never reference source,
bytecode or pickled modules.  Parity is thereby defined against torch 2.10.0 CPU.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("RECSYS_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    class _L:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    loguru = types.ModuleType("loguru")
    loguru.logger = _L()
    loguru.Logger = _L
    sys.modules["loguru"] = loguru
    sys.modules.setdefault("lmdb", types.ModuleType("lmdb"))


_install_stubs()
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from src import losses as ref_losses  # noqa: E402
from src.graph_utils import calculate_sparse_graph_adj_norm  # noqa: E402
from src.models import get_ctr_model, get_graph_model  # noqa: E402
from src.models.embeddings import get_embedding  # noqa: E402
from src.models.embeddings.dh_embedding import DHEmbedding  # noqa: E402
from src.models.layer_dcn import DCN_MixHead, DCNHead  # noqa: E402
from src.utils import set_seed  # noqa: E402


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"  wrote {name}.npz ({len(out)} arrays)")


def params_of(module, prefix="param/"):
    return {prefix + k: v.detach().clone() for k, v in module.state_dict().items()
            if isinstance(v, torch.Tensor)}


def grads_of(module, prefix="grad/"):
    return {prefix + k: p.grad.detach().clone() for k, p in module.named_parameters()
            if p.grad is not None}


def randomize_bn(module, gen):
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)


# ------------------------------------------------------------------ G1: DeepFM
def gen_deepfm():
    gen = torch.Generator().manual_seed(2023)
    cases = [
        ("deepfm_small_nobn_train", [5, 7, 11], 4, [8, 8], False, True, 6),
        ("deepfm_small_bn_train", [5, 7, 11], 4, [8, 8], True, True, 6),
        ("deepfm_small_bn_eval", [5, 7, 11], 4, [8, 8], True, False, 6),
        ("deepfm_f39_d16_bn_train", None, 16, [32, 32, 32], True, True, 64),
        ("deepfm_f26_d16_nobn_eval", None, 16, [16], False, False, 33),
    ]
    for name, dims, D, hidden, bn, training, B in cases:
        if dims is None:
            nf = 39 if "f39" in name else 26
            dims = torch.randint(2, 60, (nf,), generator=gen).tolist()
        set_seed(2023)
        cfg = {"name": "deepfm", "num_factor": D, "hidden_sizes": list(hidden), "p_dropout": 0.0,
               "use_batchnorm": bn}
        model = get_ctr_model(dims, cfg)
        with torch.no_grad():
            model._bias.copy_(torch.randn(1, generator=gen) * 0.1)
        randomize_bn(model, gen)
        model.train(training)
        x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], dim=1)
        y = (torch.rand(B, generator=gen) < 0.25).float()
        before = params_of(model)
        logits = model(x)
        loss = torch.nn.BCEWithLogitsLoss()(logits, y)
        loss.backward()
        save(name, field_dims=np.array(dims), x=x, y=y, logits=logits, loss=loss,
             use_bn=np.array(bn), training=np.array(training), n_hidden=np.array(len(hidden)),
             **before, **grads_of(model))


# ------------------------------------------------------------------ G1 / C1: the reference's base config on its own sample
def gen_deepfm_c1():
    """BASELINE config 1: the reference's OWN dataset class maps tests/assets/train_criteo_sample.txt
    (100 rows, min_threshold 10) to the 39 per-field ids and field sizes, and the `model` block of
    configs/deepfm/base_config.yaml:1-10 (D = 16, 400x3, BatchNorm; p_dropout 0 for parity) runs on B = 256 of its rows
    (the 100 rows repeated), training mode (forward + BCE backward) and eval mode.  The eval file keeps only what differs
    from the training file (`params_from`): the inputs, the BatchNorm running statistics it ran with, outputs, and the
    gradients of the small parameters."""
    import tempfile

    import yaml
    from src.dataset.criteo import CriteoDataset

    with tempfile.TemporaryDirectory() as tmp:
        ds = CriteoDataset(os.path.join(REF, "tests/assets/train_criteo_sample.txt"), os.path.join(tmp, "cache.bin"))
        dims = [int(d) for d in ds.field_dims]
        feats = torch.stack([ds[i][0] for i in range(len(ds))]).to(torch.int64)
        labels = torch.tensor([float(ds[i][1]) for i in range(len(ds))])
    assert len(dims) == 39 and feats.shape == (100, 39)
    with open(os.path.join(REF, "configs/deepfm/base_config.yaml")) as fin:
        block = yaml.safe_load(fin)["model"]
    assert block["num_factor"] == 16 and block["hidden_sizes"] == [400, 400, 400] and block["use_batchnorm"] is True
    B = 256
    pick = torch.arange(B) % feats.shape[0]
    x, y = feats[pick].contiguous(), labels[pick].contiguous()
    gen = torch.Generator().manual_seed(2024)
    for training in (True, False):
        set_seed(2023)
        cfg = dict(block, name="deepfm", p_dropout=0.0)
        model = get_ctr_model(dims, cfg)
        with torch.no_grad():
            model._bias.copy_(torch.randn(1, generator=torch.Generator().manual_seed(5)) * 0.1)
        randomize_bn(model, torch.Generator().manual_seed(6))
        model.train(training)
        before = params_of(model)
        logits = model(x)
        loss = torch.nn.BCEWithLogitsLoss()(logits, y)
        loss.backward()
        common = dict(field_dims=np.array(dims), x=x, y=y, logits=logits, loss=loss, use_bn=np.array(True),
                      training=np.array(training), n_hidden=np.array(3))
        if training:
            save("deepfm_c1_base_config_train", **common, **before, **grads_of(model))
        else:
            small = {k: v for k, v in grads_of(model).items() if v.numel() <= 4096}
            save("deepfm_c1_base_config_eval", params_from=np.array("deepfm_c1_base_config_train"), **common, **small)


# ------------------------------------------------------------------ G2: QR
def gen_qr():
    gen = torch.Generator().manual_seed(7)
    dims = [13, 29, 7]
    N, D = sum(dims), 8
    x1 = torch.randint(0, N, (11,), generator=gen)
    x2 = torch.randint(0, N, (6, 3), generator=gen)
    for op in ["mult", "add", "cat"]:
        for divider in [2, 5, None]:
            set_seed(2023)
            cfg = {"name": "qr", "operation": op}
            if divider is not None:
                cfg["divider"] = divider
            emb = get_embedding(cfg, dims, D)
            out = {}
            for tag, x in (("1d", x1), ("2d", x2)):
                emb.zero_grad()
                o = emb(x)
                G = torch.randn(o.shape, generator=gen)
                (o * G).sum().backward()
                out.update({f"x_{tag}": x, f"out_{tag}": o, f"G_{tag}": G})
                out.update(grads_of(emb, f"grad_{tag}/"))
            emb.zero_grad()
            w = emb.get_weight()
            save(f"qr_{op}_div{divider}", field_dims=np.array(dims), hidden=np.array(D),
                 divider=np.array(emb._divider), weight=w, **params_of(emb), **out)


# ------------------------------------------------------------------ G3: CERP
def gen_cerp():
    gen = torch.Generator().manual_seed(11)
    dims = [13, 29, 7]
    N, D, bucket = sum(dims), 8, 10
    x1 = torch.randint(0, N, (11,), generator=gen)
    x2 = torch.randint(0, N, (6, 3), generator=gen)
    for tag_thr in ["default", "pruning"]:
        set_seed(2023)
        emb = get_embedding({"name": "cerp", "bucket_size": bucket}, dims, D, field_name="deepfm")
        if tag_thr == "pruning":
            with torch.no_grad():
                emb.q_threshold.copy_(torch.randn(bucket, D, generator=gen) - 1.5)
                emb.p_threshold.copy_(torch.randn(bucket, D, generator=gen) - 1.5)
        out = {}
        for tag, x in (("1d", x1), ("2d", x2)):
            emb.zero_grad()
            o = emb(x)
            G = torch.randn(o.shape, generator=gen)
            (o * G).sum().backward()
            out.update({f"x_{tag}": x, f"out_{tag}": o, f"G_{tag}": G})
            out.update(grads_of(emb, f"grad_{tag}/"))
        emb.zero_grad()
        save(f"cerp_{tag_thr}", field_dims=np.array(dims), hidden=np.array(D), bucket=np.array(bucket),
             q_entity_per_row=np.array(emb.q_entity_per_row), weight=emb.get_weight(),
             num_params=np.array(emb.get_num_params()), **params_of(emb), **out)


# ------------------------------------------------------------------ G3b: PEP / PEP-retrain
def gen_pep():
    import tempfile

    gen = torch.Generator().manual_seed(31)
    dims = [13, 29, 7]
    N, D = sum(dims), 8
    x1 = torch.randint(0, N, (11,), generator=gen)
    x2 = torch.randint(0, N, (6, 3), generator=gen)
    for ttype in ["global", "dimension", "feature", "feature_dim"]:
        with tempfile.TemporaryDirectory() as tmp:
            set_seed(2023)
            emb = get_embedding({"name": "pep", "threshold_type": ttype, "checkpoint_weight_dir": tmp}, dims, D,
                                field_name="deepfm")
            with torch.no_grad():
                emb.s.copy_(torch.randn(emb.s.shape, generator=gen) * 0.7 - 1.8)
            out = {}
            for tag, x in (("1d", x1), ("2d", x2)):
                emb.zero_grad()
                o = emb(x)
                G = torch.randn(o.shape, generator=gen)
                (o * G).sum().backward()
                out.update({f"x_{tag}": x, f"out_{tag}": o, f"G_{tag}": G})
                out.update(grads_of(emb, f"grad_{tag}/"))
            emb.zero_grad()
            save(f"pep_{ttype}", field_dims=np.array(dims), hidden=np.array(D), weight=emb.get_weight(),
                 num_params=np.array(emb.get_num_params()), **params_of(emb), **out)
            if ttype == "feature_dim":
                # retrain variant: mask from this checkpoint, fresh trainable weights
                os.makedirs(os.path.join(tmp, "deepfm"), exist_ok=True)
                torch.save(emb.state_dict(), os.path.join(tmp, "deepfm", "0.5.pth"))
                set_seed(7)
                re = get_embedding({"name": "pep_retrain", "checkpoint_weight_dir": tmp, "sparsity": 0.5}, dims, D,
                                   field_name="deepfm")
                o = re(x2)
                G = torch.randn(o.shape, generator=gen)
                (o * G).sum().backward()
                save("pep_retrain", field_dims=np.array(dims), hidden=np.array(D), x=x2, out=o, G=G,
                     ckpt_weight=emb.emb.weight, ckpt_s=emb.s, weight=re.get_weight(), nnz=np.array(int(re.get_num_params())),
                     **params_of(re), **grads_of(re))


# ------------------------------------------------------------------ G3c: PTQ (fp16 / int8 / int16 tables)
def gen_ptq():
    import tempfile

    from src.models.embeddings.ptq_emb import PTQEmb_Fp16, PTQEmb_Int

    gen = torch.Generator().manual_seed(37)
    N, D = 97, 16
    W = (torch.rand(N, D, generator=gen) - 0.5) * 0.2
    x = torch.randint(0, N, (9, 4), generator=gen)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "c.pth")
        torch.save({"state_dict": {"embedding._emb_module.weight": W}}, path)
        out = {"W": W, "x": x}
        e = PTQEmb_Fp16(None, None, None, path)
        out["fp16_weight"] = e.weight.view(torch.int16)      # raw half bits
        out["fp16_out"] = e(x)
        for bits in (8, 16):
            e = PTQEmb_Int(None, None, None, path, n_bits=bits)
            out[f"int{bits}_weight"] = e.weight
            out[f"int{bits}_scale"] = e.scale
            out[f"int{bits}_bias"] = e.bias
            out[f"int{bits}_out"] = e(x)
            out[f"int{bits}_full"] = e.get_weight()
        save("ptq", **out)


# ------------------------------------------------------------------ G3d: QAT (stochastic rounding)
def gen_qat():
    """QAT_EmbInt forward/backward with the torch.rand_like draw captured: the forward consumes exactly one
    rand_like of the gathered rows' shape, so re-seeding and drawing the same shape reproduces it."""
    from src.models.embeddings.qat_emb import QAT_EmbInt

    gen = torch.Generator().manual_seed(41)
    dims = [13, 29, 7]
    N, D = sum(dims), 8
    x = torch.randint(0, N, (6, 3), generator=gen)
    for bits in (8, 16):
        set_seed(2023)
        emb = QAT_EmbInt(dims, D, None, n_bits=bits)
        with torch.no_grad():
            # push a few weights beyond the clamps so that both saturation branches of the backward are hit
            emb._emb_module.weight[x[0, 0]] *= 40.0
            emb._emb_module.weight[x[1, 1]] *= -40.0
        torch.manual_seed(99)
        o = emb(x)
        torch.manual_seed(99)
        prob = torch.rand(6, 3, D)
        G = torch.randn(o.shape, generator=gen)
        (o * G).sum().backward()
        save(f"qat_int{bits}", field_dims=np.array(dims), hidden=np.array(D), n_bits=np.array(bits), x=x, prob=prob,
             out=o, G=G, **params_of(emb), **grads_of(emb))


# ------------------------------------------------------------------ G3e: OptEmbed supernet lookup (DeepFM variant)
def gen_optembed():
    """Training forward/backward with the torch.randint dimension-mask draw captured (first RNG use of the forward),
    thresholds placed near the row norms so that BinaryStep's surrogate gradient is exercised; eval get_weight with
    and without a dimension mask for field / feature thresholds."""
    from src.models.embeddings.deepfm_opt_embed import OptEmbed

    gen = torch.Generator().manual_seed(43)
    dims = [13, 29, 7]
    N, D, B = sum(dims), 8, 6
    offs = torch.tensor([0, 13, 42])
    x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1) + offs
    for norm in (1, 2):
        set_seed(2023)
        emb = OptEmbed(dims, D, None, t_init=0.0, mode_threshold_e="field", mode_threshold_d="field", norm=norm)
        with torch.no_grad():
            typical = torch.norm(emb._weight, norm, dim=1).mean()
            emb._mask_e_module._t_param.copy_(typical + torch.tensor([-0.3, 0.05, 0.6]) * typical)
        emb.train()
        torch.manual_seed(77)
        o = emb(x)
        torch.manual_seed(77)
        mask_d = torch.randint(0, D, size=(B, len(dims)))
        G = torch.randn(o.shape, generator=gen)
        (o * G).sum().backward()
        out = dict(field_dims=np.array(dims), hidden=np.array(D), norm=np.array(norm), x=x, mask_d=mask_d, out=o, G=G,
                   l_s=emb.get_l_s(), **params_of(emb), **grads_of(emb))
        emb.zero_grad()
        emb.eval()
        out["weight_eval"] = emb.get_weight().clone()
        md = torch.randint(0, D, (len(dims),), generator=gen)
        out["mask_d_field"] = md
        out["weight_eval_masked"] = emb.get_weight(md).clone()
        out["lookup_eval_masked"] = emb(x).clone()
        sp, nnz = emb.get_sparsity(True)
        out["nnz"] = np.array(nnz)
        save(f"optembed_l{norm}", **out)
    # per-feature thresholds and per-feature dimension masks (eval only: the training forward asserts field thresholds)
    set_seed(7)
    emb = OptEmbed(dims, D, None, t_init=0.0, mode_threshold_e="feature", mode_threshold_d="feature", norm=1)
    with torch.no_grad():
        nrm = torch.norm(emb._weight, 1, dim=1)
        emb._mask_e_module._t_param.copy_(nrm + (torch.rand(N, generator=gen) - 0.5))
    emb.eval()
    md = torch.randint(0, D, (N,), generator=gen)
    save("optembed_feature", field_dims=np.array(dims), hidden=np.array(D), x=x, mask_d=md,
         weight_eval=emb.get_weight().clone(), weight_eval_masked=emb.get_weight(md).clone(),
         lookup_eval_masked=emb(x).clone(), **params_of(emb))


# ------------------------------------------------------------------ G4: DHE
def gen_dhe():
    gen = torch.Generator().manual_seed(13)
    for name, dims, k, hidden, D in [("dhe_k64", [23, 31], 64, [32, 32], 16),
                                     ("dhe_k1024", [1000], 1024, [64], 16)]:
        DHEmbedding.COUNTER = 5 if name == "dhe_k64" else 0
        set_seed(2023)
        emb = DHEmbedding(dims, D, None, k, list(hidden), cached=False)
        randomize_bn(emb, gen)
        emb.eval()
        N = sum(dims)
        ids = torch.cat([torch.tensor([0, 1, N - 1]), torch.randint(0, N, (29,), generator=gen)])
        big = torch.tensor([0, 123456789, 10**9, 33762576, 2**31 + 5])
        h = emb._get_universal_hash_batch(ids)
        hbig = emb._get_universal_hash_batch(big)
        with torch.no_grad():
            out = emb(ids)
            x2 = ids[:30].reshape(10, 3)
            out2 = emb(x2)
        save(name, field_dims=np.array(dims), k=np.array(k), hidden=np.array(hidden), D=np.array(D),
             prefix=np.array(emb._prefix), slopes=emb._slopes, bias=emb._bias,
             primes=emb._primes_choices, ids=ids, hash=h, big_ids=big, big_hash=hbig,
             out=out, x2=x2, out2=out2, **params_of(emb._seq, "param/_seq."))
    # per-item hash family (use_universal_hash=False): the host-built [N, k] feature table and both lookup flows
    DHEmbedding.COUNTER = 3
    set_seed(2023)
    dims, k, hidden, D = [7, 9], 16, [8], 4
    emb = DHEmbedding(dims, D, None, k, list(hidden), cached=True, use_universal_hash=False)
    after_init = torch.randint(0, 2**31 - 1, (4,))          # where the constructor leaves the global generator
    cache = emb._cache.cpu() if isinstance(emb._cache, torch.Tensor) else torch.stack(emb._cache)
    randomize_bn(emb, gen)
    emb.eval()
    N = sum(dims)
    ids = torch.cat([torch.tensor([0, 1, N - 1]), torch.randint(0, N, (13,), generator=gen)])
    x2 = ids[:12].reshape(4, 3)
    with torch.no_grad():
        out, out2 = emb(ids), emb(x2)
    DHEmbedding.COUNTER = 3
    set_seed(2023)
    lazy = DHEmbedding(dims, D, None, k, list(hidden), cached=False, use_universal_hash=False)
    lazy.load_state_dict(emb.state_dict())
    lazy.eval()
    with torch.no_grad():
        out_lazy = lazy(x2)
    save("dhe_peritem", field_dims=np.array(dims), k=np.array(k), hidden=np.array(hidden), D=np.array(D),
         prefix=np.array(emb._prefix), cache=cache, after_init=after_init, ids=ids, out=out, x2=x2, out2=out2,
         out_lazy=out_lazy, **params_of(emb._seq, "param/_seq."))
    DHEmbedding.COUNTER = 0


# ------------------------------------------------------------------ G5: TT-Rec (torch)
def gen_tt():
    gen = torch.Generator().manual_seed(17)
    for name, N, D, ranks, p_shapes, q_shapes in [
        ("tt_r3x3", 512, 16, [3, 3], None, None),
        ("tt_r2x4x2", 600, 16, [2, 4, 2], [5, 5, 5, 5], [2, 2, 2, 2]),
        ("tt_r4", 100, 8, [4], [10, 10], [2, 4]),
    ]:
        set_seed(2023)
        cfg = {"name": "tt_emb_torch", "tt_ranks": list(ranks)}
        if p_shapes is not None:
            cfg["tt_p_shapes"], cfg["tt_q_shapes"] = list(p_shapes), list(q_shapes)
        emb = get_embedding(cfg, N, D)
        x1 = torch.cat([torch.arange(10), torch.randint(0, N, (23,), generator=gen)])
        x2 = torch.randint(0, N, (5, 4), generator=gen)
        out = {}
        for tag, x in (("1d", x1), ("2d", x2)):
            emb.zero_grad()
            o = emb(x)
            G = torch.randn(o.shape, generator=gen)
            (o * G).sum().backward()
            out.update({f"x_{tag}": x, f"out_{tag}": o, f"G_{tag}": G})
            out.update(grads_of(emb, f"grad_{tag}/"))
        emb.zero_grad()
        save(name, num_item=np.array(N), hidden=np.array(D), tt_ranks=np.array(emb.tt_ranks),
             tt_p_shapes=np.array(emb.tt_p_shapes), tt_q_shapes=np.array(emb.tt_q_shapes),
             weight=emb.get_weight(), **params_of(emb), **out)


def gen_tt_init():
    """weight_dist='approx-uniform': the seeded construction itself is the fixture (numpy + Python `random` streams)."""
    gen = torch.Generator().manual_seed(171)
    for name, N, D, ranks, p_shapes, q_shapes in [
        ("ttinit_approx_uniform_r4x6", 512, 16, [4, 6], None, None),
        ("ttinit_approx_uniform_r2x3", 90, 8, [2, 3], [3, 5, 6], [2, 2, 2]),
    ]:
        set_seed(2023)
        cfg = {"name": "tt_emb_torch", "tt_ranks": list(ranks), "weight_dist": "approx-uniform"}
        if p_shapes is not None:
            cfg["tt_p_shapes"], cfg["tt_q_shapes"] = list(p_shapes), list(q_shapes)
        emb = get_embedding(cfg, N, D)
        x = torch.randint(0, N, (6, 3), generator=gen)
        save(name, num_item=np.array(N), hidden=np.array(D), tt_ranks=np.array(emb.tt_ranks),
             tt_p_shapes=np.array(emb.tt_p_shapes), tt_q_shapes=np.array(emb.tt_q_shapes), x=x, out=emb(x),
             weight=emb.get_weight(), **params_of(emb))


# ------------------------------------------------------------------ G6: DCN
def gen_dcn():
    gen = torch.Generator().manual_seed(19)
    # heads alone
    set_seed(2023)
    head = DCN_MixHead(num_experts=4, num_layers=3, rank=8, hidden_size=24)
    with torch.no_grad():
        for b in head.biases:
            b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
    x0 = torch.randn(10, 24, generator=gen, requires_grad=True)
    o = head(x0 * 0.5)
    G = torch.randn(o.shape, generator=gen)
    (o * G).sum().backward()
    save("dcn_mixhead", x=x0, out=o, G=G, grad_x=x0.grad, **params_of(head), **grads_of(head))

    set_seed(2023)
    head = DCNHead(3, 24)
    x0 = torch.randn(10, 24, generator=gen, requires_grad=True)
    o = head(x0)
    G = torch.randn(o.shape, generator=gen)
    (o * G).sum().backward()
    save("dcn_head", x=x0, out=o, G=G, grad_x=x0.grad, **params_of(head), **grads_of(head))

    # full models, dropout off, train mode (batch-stat BN) and eval
    dims = [5, 7, 11, 3]
    for name, cfg in [
        ("dcn_mix_vanilla", {"name": "dcn_mix", "num_factor": 4, "hidden_sizes": [8, 8], "num_layers": 2,
                             "num_experts": 3, "rank": 5, "p_dropout": 0.0, "compile_model": False}),
        ("dcn_mix_qr", {"name": "dcn_mix", "num_factor": 4, "hidden_sizes": [8], "num_layers": 3,
                        "num_experts": 4, "rank": 6, "p_dropout": 0.0, "compile_model": False,
                        "embedding_config": {"name": "qr", "divider": 2, "operation": "mult"}}),
        ("dcnv2_stacked", {"name": "dcn", "num_factor": 4, "hidden_sizes": [8, 8], "num_layers": 2,
                           "p_dropout": 0.0, "structure": "Stacked"}),
        ("dcnv2_parallel", {"name": "dcn", "num_factor": 4, "hidden_sizes": [8], "num_layers": 3,
                            "p_dropout": 0.0, "structure": "Parallel"}),
    ]:
        for training in (True, False):
            set_seed(2023)
            model = get_ctr_model(dims, dict(cfg))
            randomize_bn(model, gen)
            model.train(training)
            B = 9
            x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], dim=1)
            y = (torch.rand(B, generator=gen) < 0.4).float()
            before = params_of(model)
            logits = model(x)
            loss = torch.nn.BCEWithLogitsLoss()(logits, y)
            loss.backward()
            save(f"{name}_{'train' if training else 'eval'}", field_dims=np.array(dims), x=x, y=y,
                 logits=logits, loss=loss, training=np.array(training), **before, **grads_of(model))


# ------------------------------------------------------------------ G7: LightGCN
def load_cf_graph(path):
    graph = {}
    num_item = 0
    with open(path) as fin:
        for line in fin.readlines():
            info = line.strip().split()
            items = [int(i) for i in info[1:]]
            if not items:
                continue
            graph[int(info[0])] = items
            num_item = max(*items, num_item)
    return graph, num_item + 1


def gen_lightgcn():
    gen = torch.Generator().manual_seed(23)
    graph, num_item = load_cf_graph(os.path.join(REF, "tests/assets/sample_cf.txt"))
    num_user = len(graph)
    eu, ei = [], []
    for u, items in graph.items():
        eu.extend([u] * len(items))
        ei.extend(items)
    adj = calculate_sparse_graph_adj_norm(graph, num_item, num_user)
    save("cf_sample_adj", edge_user=np.array(eu), edge_item=np.array(ei), num_user=np.array(num_user),
         num_item=np.array(num_item), crow=adj.crow_indices(), col=adj.col_indices(), val=adj.values())
    B = 16
    users = torch.randint(0, num_user, (B,), generator=gen)
    pos = torch.randint(0, num_item, (B,), generator=gen)
    neg = torch.randint(0, num_item, (B,), generator=gen)
    for mname in ["lightgcn", "single-lightgcn"]:
        for L in (1, 2, 3):
            set_seed(2023)
            model = get_graph_model(num_user, num_item, {"name": mname, "num_layers": L, "hidden_size": 16})
            ue, ie = model(adj)
            loss = ref_losses.bpr_loss(ue[users], ie[pos], ie[neg])
            reg = model.get_reg_loss(users, pos, neg)
            (loss + 1e-3 * reg).backward()
            save(f"{mname.replace('-', '_')}_L{L}", users=users, pos=pos, neg=neg, user_emb=ue, item_emb=ie,
                 bpr=loss, reg=reg, num_layers=np.array(L), **params_of(model), **grads_of(model))


# ------------------------------------------------------------------ G8: CSR-pruned table
def gen_csr_pruned():
    # numba is absent here, so PrunedEmbedding itself cannot run; the fixture pins the
    # equivalence the reference's own test asserts (tests/test_emb.py:375-393): lookup of a
    # 90%-sparse table through torch's CSR form == dense nn.Embedding lookup.
    gen = torch.Generator().manual_seed(29)
    N, D = 1024, 16
    nnz = int(N * D * 0.1)
    ind = torch.stack([torch.randint(N, (nnz,), generator=gen), torch.randint(D, (nnz,), generator=gen)])
    dense = torch.sparse_coo_tensor(ind, torch.randn(nnz, generator=gen), size=(N, D)).to_dense()
    csr = dense.to_sparse_csr()
    ids = torch.randint(N, (256,), generator=gen)
    ids2 = torch.randint(N, (16, 5), generator=gen)
    save("csr_pruned", values=csr.values(), crow=csr.crow_indices(), col=csr.col_indices(),
         dense=dense, ids=ids, out=torch.nn.functional.embedding(ids, dense), ids2=ids2,
         out2=torch.nn.functional.embedding(ids2, dense))


# ------------------------------------------------------------------ G9: the two losses around the BPR loss
def gen_losses():
    gen = torch.Generator().manual_seed(31)
    n, D, K = 37, 16, 5
    arrays = {}
    for tag, b_cos, temp in (("cos_t02", True, 0.2), ("cos_t1", True, 1.0), ("dot_t05", False, 0.5)):
        v1 = torch.randn(n, D, generator=gen).requires_grad_(True)
        v2 = torch.randn(n, D, generator=gen).requires_grad_(True)
        loss = ref_losses.info_nce(v1, v2, temp, b_cos)
        loss.backward()
        arrays.update({f"{tag}/v1": v1, f"{tag}/v2": v2, f"{tag}/temperature": np.array(temp), f"{tag}/b_cos": np.array(b_cos),
                       f"{tag}/loss": loss, f"{tag}/grad_v1": v1.grad, f"{tag}/grad_v2": v2.grad})
    # the trainer's call: both views are the same matrix (src/trainer/lightgcn.py:219-227)
    v = torch.randn(n, D, generator=gen).requires_grad_(True)
    loss = ref_losses.info_nce(v, v, 0.2)
    loss.backward()
    arrays.update({"self/v": v, "self/loss": loss, "self/grad_v": v.grad})
    u = torch.randn(n, D, generator=gen).requires_grad_(True)
    p = torch.randn(n, D, generator=gen).requires_grad_(True)
    ng = torch.randn(n, K, D, generator=gen).requires_grad_(True)
    loss = ref_losses.bpr_loss_multi(u, p, ng)
    loss.backward()
    arrays.update({"multi/u": u, "multi/p": p, "multi/n": ng, "multi/loss": loss, "multi/grad_u": u.grad,
                   "multi/grad_p": p.grad, "multi/grad_n": ng.grad})
    save("losses", **arrays)


# ------------------------------------------------------------------ G10: ranking metrics of the LightGCN validation
def gen_metrics():
    from src import metrics as ref_metrics

    gen = torch.Generator().manual_seed(37)
    n, k, num_item = 50, 20, 400
    pred = torch.stack([torch.randperm(num_item, generator=gen)[:k] for _ in range(n)])
    lens = torch.randint(1, 45, (n,), generator=gen)
    true = [torch.randperm(num_item, generator=gen)[:int(m)].tolist() for m in lens]
    true[3] = pred[3, :5].tolist()                      # a user whose whole test set is recommended
    true_pad = torch.full((n, int(lens.max())), -1, dtype=torch.int64)
    for i, t in enumerate(true):
        true_pad[i, :len(t)] = torch.tensor(t)
    ndcg, recall = ref_metrics.get_ndcg_recall(pred.tolist(), [set(t) for t in true], k)
    ndcg_only = ref_metrics.get_ndcg(pred.tolist(), true, k)
    ndcg10, recall10 = ref_metrics.get_ndcg_recall(pred.tolist(), true, 10)
    save("metrics", pred=pred, true_pad=true_pad, k=np.array(k), ndcg=np.array(ndcg), recall=np.array(recall),
         ndcg_only=np.array(ndcg_only), ndcg10=np.array(ndcg10), recall10=np.array(recall10))


if __name__ == "__main__":
    torch.set_num_threads(1)
    which = sys.argv[1:] or ["deepfm", "deepfm_c1", "qr", "cerp", "pep", "ptq", "qat", "optembed", "dhe", "tt", "tt_init", "dcn", "lightgcn", "csr_pruned", "losses", "metrics"]
    for w in which:
        print(f"[{w}]")
        globals()[f"gen_{w}"]()
