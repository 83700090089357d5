"""ORACLE — test infrastructure, NOT product code.

CPU restatement of the reference's arithmetic for the hot path,
written as pure functions over plain tensors in stock PyTorch CPU ops — the same
op sequence the reference executes (the reference IS stock PyTorch on CPU), so
autograd yields the reference's gradients.  Each function cites the reference
file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (recsys-benchmark_amd/)
never does and fails loudly without its HIP library.

Pinned (tests/test_oracle_golden.py) against golden vectors generated in the
build container by importing the reference itself (tests/golden/gen_golden.py),
parity being defined against torch 2.10.0 CPU as SURVEY.md §8c states.
Parameters are passed as dicts keyed by the reference's state_dict names.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- DeepFM
def field_offsets(field_dims: Sequence[int]) -> torch.Tensor:
    """offsets = cumsum([0] + field_dims[:-1]), shape [1,F] int64 (src/models/deepfm.py:71-76)."""
    t = torch.cat([torch.tensor([0], dtype=torch.long), torch.tensor(list(field_dims))])
    return torch.cumsum(t[:-1], 0).unsqueeze(0)


def fm_second_order(emb: torch.Tensor) -> torch.Tensor:
    """0.5 * sum_d[(sum_f e)^2 - sum_f e^2], [B,F,D] -> [B,1] (src/models/deepfm.py:91-92,98)."""
    square_of_sum = emb.sum(dim=1).pow(2)
    sum_of_square = emb.pow(2).sum(dim=1)
    return 0.5 * (square_of_sum - sum_of_square).sum(1, keepdim=True)


def first_order(rows: torch.Tensor, fc_weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """EmbeddingBag(N,1,'sum') over each row of the 2-D input + bias (src/models/deepfm.py:49,95)."""
    return F.embedding_bag(rows, fc_weight, mode="sum") + bias


def mlp_tail(x: torch.Tensor, p: Params, prefix: str, hidden: int, use_bn: bool, training: bool,
             bn_eps: float = 1e-5, p_dropout: float = 0.0) -> torch.Tensor:
    """(Linear, [BatchNorm1d], ReLU, Dropout(p=0 here))xk + Linear(.,1)
    (src/models/deepfm.py:53-66; src/models/dcn.py:56-66).  Dropout must be off for parity
    (SURVEY.md §7 'BatchNorm + Dropout').  BatchNorm uses batch statistics when training
    (running stats are not updated here: the oracle is functional)."""
    i = 0
    step = 4 if use_bn else 3
    for _ in range(hidden):
        x = F.linear(x, p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"])
        if use_bn:
            x = F.batch_norm(
                x, p[f"{prefix}.{i+1}.running_mean"].clone(), p[f"{prefix}.{i+1}.running_var"].clone(),
                p[f"{prefix}.{i+1}.weight"], p[f"{prefix}.{i+1}.bias"], training=training, eps=bn_eps,
            )
        x = F.relu(x)
        if p_dropout > 0.0:  # only the cpu_baseline timing leg uses this (RNG-dependent)
            x = F.dropout(x, p_dropout, training)
        i += step
    return F.linear(x, p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"])


def deepfm_embed_fm(x: torch.Tensor, p: Params) -> Tuple[torch.Tensor, torch.Tensor]:
    """The gather + FM + first-order part of DeepFM.forward with a vanilla table
    (src/models/deepfm.py:88-98; src/models/embeddings/base.py:74-75) -> (emb[B,F,D], y_fm[B,1])."""
    rows = x + p["offsets"]
    emb = F.embedding(rows, p["embedding._emb_module.weight"])
    y_fm = first_order(rows, p["fc.weight"], p["_bias"]) + fm_second_order(emb)
    return emb, y_fm


def deepfm_forward(x: torch.Tensor, p: Params, n_hidden: int, use_bn: bool, training: bool,
                   p_dropout: float = 0.0) -> torch.Tensor:
    """DeepFM.forward, vanilla embedding (src/models/deepfm.py:79-105); dropout off unless asked."""
    emb, y_fm = deepfm_embed_fm(x, p)
    b, nf, d = emb.shape
    scores = y_fm + mlp_tail(emb.reshape(b, nf * d), p, "_deep_branch", n_hidden, use_bn, training,
                             p_dropout=p_dropout)
    return scores.squeeze(-1)
