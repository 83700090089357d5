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


def first_order(rows: torch.Tensor, fc_weight: torch.Tensor, bias: torch.Tensor, sparse: bool = False) -> torch.Tensor:
    """EmbeddingBag(N,1,'sum') over each row of the 2-D input + bias (src/models/deepfm.py:49,95)."""
    return F.embedding_bag(rows, fc_weight, mode="sum", sparse=sparse) + bias


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


def deepfm_embed_fm(x: torch.Tensor, p: Params, sparse: bool = False, fc_sparse: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """The gather + FM + first-order part of DeepFM.forward with a vanilla table
    (src/models/deepfm.py:88-98; src/models/embeddings/base.py:74-75) -> (emb[B,F,D], y_fm[B,1]).
    sparse: nn.Embedding(sparse=True), the `sparse: True` embedding config of configs/deepfm/base_config_sparse.yaml:8-10
    (src/models/embeddings/base.py:60-64) — same values, the table's gradient in row (COO) form; fc_sparse: the same for
    the first-order EmbeddingBag (this build's extension, DeepFM(fc_sparse=True))."""
    rows = x + p["offsets"]
    emb = F.embedding(rows, p["embedding._emb_module.weight"], sparse=sparse)
    y_fm = first_order(rows, p["fc.weight"], p["_bias"], sparse=fc_sparse) + fm_second_order(emb)
    return emb, y_fm


def deepfm_forward(x: torch.Tensor, p: Params, n_hidden: int, use_bn: bool, training: bool,
                   p_dropout: float = 0.0, sparse: bool = False, fc_sparse: bool = False) -> torch.Tensor:
    """DeepFM.forward, vanilla embedding (src/models/deepfm.py:79-105); dropout off unless asked."""
    emb, y_fm = deepfm_embed_fm(x, p, sparse, fc_sparse)
    b, nf, d = emb.shape
    scores = y_fm + mlp_tail(emb.reshape(b, nf * d), p, "_deep_branch", n_hidden, use_bn, training,
                             p_dropout=p_dropout)
    return scores.squeeze(-1)


# --------------------------------------------------------------------------- QR hashing
def qr_table_sizes(num_item: int, divider: Optional[int]) -> Tuple[int, int, int]:
    """(divider, rows of emb1, rows of emb2) — src/models/embeddings/qr_embedding.py:46-56."""
    if divider is None:
        divider = int(math.sqrt(num_item))
    return divider, divider, (num_item - 1) // divider + 1


def qr_forward(idx: torch.Tensor, emb1: torch.Tensor, emb2: torch.Tensor, divider: int, operation: str):
    """src/models/embeddings/qr_embedding.py:95-109 (mode=None): emb1 by remainder, emb2 by
    quotient; 'cat' concatenates on dim=1 (so [B,F] ids -> [B,2F,D/2])."""
    e1 = F.embedding(idx % divider, emb1)
    e2 = F.embedding(idx // divider, emb2)
    if operation == "cat":
        return torch.cat([e1, e2], dim=1)
    if operation == "add":
        return e1 + e2
    if operation == "mult":
        return e1 * e2
    raise NotImplementedError(operation)


# --------------------------------------------------------------------------- CERP
def soft_threshold(w: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """sign(w) * relu(|w| - sigmoid(s)) — src/models/embeddings/cerp_embedding.py:142-148."""
    return torch.sign(w) * torch.relu(torch.abs(w) - torch.sigmoid(s))


def cerp_forward(idx, p_weight, q_weight, p_threshold, q_threshold, bucket_size: int, q_entity_per_row: int):
    """src/models/embeddings/cerp_embedding.py:150-162 (mode=None)."""
    q_idx = torch.div(idx, q_entity_per_row, rounding_mode="trunc")
    p_idx = idx % bucket_size
    return F.embedding(q_idx, soft_threshold(q_weight, q_threshold)) + F.embedding(
        p_idx, soft_threshold(p_weight, p_threshold))


def cerp_retrain_forward(idx, p_weight, q_weight, p_mask, q_mask, bucket_size: int, q_entity_per_row: int):
    """src/models/embeddings/cerp_embedding.py:329-352 (mode=None): fixed boolean masks."""
    q_idx = torch.div(idx, q_entity_per_row, rounding_mode="trunc")
    p_idx = idx % bucket_size
    return F.embedding(q_idx, q_weight * q_mask) + F.embedding(p_idx, p_weight * p_mask)


def pep_forward(idx, weight, s) -> torch.Tensor:
    """PepEmbeeding.forward (mode=None) — src/models/embeddings/pep_embedding.py:82-92: soft-threshold the
    whole table (s broadcasts over it), then gather."""
    return F.embedding(idx, soft_threshold(weight, s))


def pep_retrain_forward(idx, weight, mask) -> torch.Tensor:
    """RetrainPepEmbedding.forward — pep_embedding.py:215-221."""
    return F.embedding(idx, weight * mask)


# --------------------------------------------------------------------------- DHE
def dhe_hash(ids: torch.Tensor, slopes, bias, primes, prefix: int, m: int = 1000000) -> torch.Tensor:
    """_get_universal_hash_batch — src/models/embeddings/dh_embedding.py:213-236.  torch's % on
    int64 is floor-mod; int / int is true division in fp32."""
    result = slopes.unsqueeze(0) * (ids.unsqueeze(1) + prefix + 1) + bias.unsqueeze(0)
    result = result % primes.unsqueeze(0) % m
    result = result / (m - 1)
    return result * 2 - 1


def dhe_item_hash(items, primes: torch.Tensor, k: int, prefix: int, m: int = 1000000) -> torch.Tensor:
    """_get_hash with use_universal_hash=False — src/models/embeddings/dh_embedding.py:155-196: per item the GLOBAL torch
    generator is re-seeded with item + prefix, then k slopes, k offsets (zeros re-drawn) and k prime picks are drawn.
    Returns [len(items), k] fp32; leaves the global generator where the last item's draws end."""
    lo, hi = -int(1e9), int(1e9)
    rows = []
    for item in items:
        torch.manual_seed(int(item) + prefix)
        a = torch.randint(lo, hi, (k,))
        b = torch.randint(lo, hi, (k,))
        while int((b == 0).sum()) > 0:
            b[b == 0] = torch.randint(lo, hi, (int((b == 0).sum()),))
        pick = primes[torch.randint(0, len(primes), (k,))]
        rows.append(((a * (int(item) + 1) + b) % pick % m) / (m - 1) * 2 - 1)
    return torch.stack(rows)


def dhe_mlp(x: torch.Tensor, p: Params, n_layers: int, use_bn: int, training: bool, prefix: str = "_seq"):
    """The Linear / BatchNorm1d / Mish stack of DHEmbedding (dh_embedding.py:99-117)."""
    i = 0
    for _ in range(n_layers):
        x = F.linear(x, p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"])
        i += 1
        if use_bn == 2:
            x = F.batch_norm(x, p[f"{prefix}.{i}.running_mean"].clone(), p[f"{prefix}.{i}.running_var"].clone(),
                             p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"], training=training)
            x = F.mish(x)
            i += 2
        elif use_bn == 1:
            x = F.mish(x)
            x = F.batch_norm(x, p[f"{prefix}.{i+1}.running_mean"].clone(), p[f"{prefix}.{i+1}.running_var"].clone(),
                             p[f"{prefix}.{i+1}.weight"], p[f"{prefix}.{i+1}.bias"], training=training)
            i += 2
        else:
            x = F.mish(x)
            i += 1
    return x


# --------------------------------------------------------------------------- CSR-pruned rows
def csr_rows(values, crow, col, ids, hidden_size: int) -> torch.Tensor:
    """csr_embedding_lookup_cpu — src/models/embeddings/pruned_embedding.py:187-204 (numba there;
    numba is not installed here, so this is pinned by the dense-lookup equivalence the reference's own
    test asserts, tests/test_emb.py:375-393).  Pure-Python loops: small cases only."""
    flat = ids.flatten().tolist()
    out = torch.zeros((len(flat), hidden_size), dtype=torch.float32)
    for k, rowid in enumerate(flat):
        for i in range(int(crow[rowid]), int(crow[rowid + 1])):
            out[k, int(col[i])] = values[i]
    return out.reshape(*ids.shape, hidden_size)


# --------------------------------------------------------------------------- LightGCN
def adj_norm_csr(edge_user, edge_item, num_user: int, num_item: int) -> torch.Tensor:
    """calculate_sparse_graph_adj_norm — src/graph_utils.py:47-98, from the (user,item) edge list in
    file order (per user: all R entries, then all R^T entries, as the reference's loop emits them).
    Degrees are column sums of the UN-coalesced COO, so a duplicate edge counts twice and its two
    normalised entries are summed by to_sparse_csr()."""
    eu = [int(u) for u in edge_user]
    ei = [int(i) for i in edge_item]
    rows: List[int] = []
    cols: List[int] = []
    k = 0
    while k < len(eu):
        j = k
        while j < len(eu) and eu[j] == eu[k]:
            j += 1
        items = [it + num_user for it in ei[k:j]]
        rows.extend([eu[k]] * len(items))
        cols.extend(items)
        cols.extend([eu[k]] * len(items))
        rows.extend(items)
        k = j
    idx = torch.tensor([rows, cols])
    n = num_user + num_item
    adj = torch.sparse_coo_tensor(idx, torch.ones(len(rows)), size=(n, n))
    degree = adj.sum(dim=0).pow(-0.5)
    values = torch.index_select(degree, 0, idx[0]) * torch.index_select(degree, 0, idx[1])
    return torch.sparse_coo_tensor(idx, values.coalesce().values(), size=(n, n)).to_sparse_csr()


def lightgcn_propagate(matrix: torch.Tensor, embs: torch.Tensor, num_layers: int) -> torch.Tensor:
    """res = mean_{k=0..L} A^k E0 — src/models/lightgcn.py:79-87 (same op order: running sum, one divide)."""
    res = embs
    step = embs
    for _ in range(num_layers):
        step = matrix @ step
        res = res + step
    return res / (num_layers + 1)


def bpr_loss(user_embs, pos_embs, neg_embs) -> torch.Tensor:
    """src/losses.py:6-22."""
    y_pos = (user_embs * pos_embs).sum(1)
    y_neg = (user_embs * neg_embs).sum(1)
    return -F.logsigmoid(y_pos - y_neg).mean()


def info_nce(view1: torch.Tensor, view2: torch.Tensor, temperature: float = 1, b_cos: bool = True) -> torch.Tensor:
    """src/losses.py:25-47: -mean_i log softmax_j(<v1_i, v2_j> / T)[i, i]; with b_cos the rows are first scaled to unit
    length the way F.normalize does it (x / max(|x|_2, 1e-12))."""
    if b_cos:
        # vector_norm, not sqrt(sum(x^2)): its backward is 0 (not NaN) at an all-zero row, as in F.normalize
        view1 = view1 / torch.linalg.vector_norm(view1, 2, 1, keepdim=True).clamp_min(1e-12)
        view2 = view2 / torch.linalg.vector_norm(view2, 2, 1, keepdim=True).clamp_min(1e-12)
    scores = view1 @ view2.T / temperature
    return (torch.logsumexp(scores, 1) - scores.diagonal()).mean()


def bpr_loss_multi(user_embs, pos_embs, neg_embs) -> torch.Tensor:
    """src/losses.py:50-68: K negatives per positive; summed over the negatives, averaged over the N samples."""
    y_pos = (user_embs * pos_embs).sum(1, keepdim=True)                 # [N, 1]
    y_neg = (user_embs.unsqueeze(1) * neg_embs).sum(2)                  # [N, K]
    return F.softplus(y_neg - y_pos).sum() / user_embs.shape[0]


def l2_reg_loss(user_rows, pos_rows, neg_rows) -> torch.Tensor:
    """LightGCN.get_reg_loss — src/models/lightgcn.py:90-100."""
    return (user_rows.norm(2).pow(2) + pos_rows.norm(2).pow(2) + neg_rows.norm(2).pow(2)) / (2 * len(user_rows))


# --------------------------------------------------------------------------- CrossNet heads
def dcn_head(x0: torch.Tensor, p: Params, num_layers: int, prefix: str = "layers") -> torch.Tensor:
    """DCNHead.forward — src/models/layer_dcn.py:129-140: x_{l+1} = x_l + x_0 * Linear_l(x_l)."""
    x_l = x0
    for l in range(num_layers):
        x_l = x_l + x0 * F.linear(x_l, p[f"{prefix}.{l}.weight"], p[f"{prefix}.{l}.bias"])
    return x_l


def dcn_mix_head(x0: torch.Tensor, p: Params, num_layers: int, prefix: str = "") -> torch.Tensor:
    """DCN_MixHead.forward + forward_mixture — src/models/layer_dcn.py:8-24,90-115 (tanh, identity gate)."""
    x_l = x0
    x0u = x0.unsqueeze(1)
    for l in range(num_layers):
        C, V, U, b = p[f"{prefix}C.{l}"], p[f"{prefix}V.{l}"], p[f"{prefix}U.{l}"], p[f"{prefix}biases.{l}"]
        E_i = torch.tanh(x_l @ V)                              # [E, B, r]
        E_i = E_i.permute(1, 0, 2)                             # [B, E, r]
        E_i = torch.tanh(torch.einsum("ber,erq->beq", E_i, C))
        E_i = torch.einsum("ber,erd->bed", E_i, U)             # [B, E, d]
        E_i = x0u * (E_i + b)
        gates = (x_l @ p[f"{prefix}gates"]).squeeze(2).permute(1, 0)   # [B, E]
        x_l = torch.einsum("be,bed->bd", gates, E_i) + x_l
    return x_l


def bn_mlp(x, p: Params, prefix: str, n_hidden: int, training: bool, relu_keep=None, pre_out=None) -> torch.Tensor:
    """(Linear, BatchNorm1d, ReLU, Dropout(0)) x n — the `_dnn` stacks of src/models/dcn.py:56-66,179-186.

    Test hooks (not part of the reference): `pre_out` (a list) receives, per layer, (z, s): the pre-activation entering the
    ReLU and the sum of the magnitudes of the terms it was formed from (what a float32 evaluation's rounding error scales
    with); `relu_keep` (a list of bool tensors) replaces the decisions `z > 0` of the ReLUs by given ones — a float32
    implementation takes the other side of the kink where z is within rounding of 0, and the comparison must hold the
    decisions fixed to see anything else."""
    for k in range(n_hidden):
        i = 4 * k
        W, b = p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"]
        lin = F.linear(x, W, b)
        z = F.batch_norm(lin, p[f"{prefix}.{i+1}.running_mean"].clone(), p[f"{prefix}.{i+1}.running_var"].clone(),
                         p[f"{prefix}.{i+1}.weight"], p[f"{prefix}.{i+1}.bias"], training=training)
        if pre_out is not None:
            with torch.no_grad():
                mean = lin.mean(0) if training else p[f"{prefix}.{i+1}.running_mean"]
                var = lin.var(0, unbiased=False) if training else p[f"{prefix}.{i+1}.running_var"]
                scale = (p[f"{prefix}.{i+1}.weight"] * torch.rsqrt(var + 1e-5)).abs()
                s_lin = F.linear(x.abs(), W.abs(), b.abs())
                pre_out.append((z.detach(), (s_lin + mean.abs()) * scale + p[f"{prefix}.{i+1}.bias"].abs()))
        x = F.relu(z) if relu_keep is None else z * relu_keep[k].to(z.dtype)
    return x


def dcn_mix_forward(x, p: Params, emb: torch.Tensor, num_layers: int, n_hidden: int, training: bool, relu_keep=None,
                    pre_out=None) -> torch.Tensor:
    """DCN_Mix.forward after the embedding lookup — src/models/dcn.py:89-96 (relu_keep / pre_out: see bn_mlp)."""
    h = dcn_mix_head(emb.reshape(x.shape[0], -1), p, num_layers, "cross_head.")
    h = bn_mlp(h, p, "_dnn", n_hidden, training, relu_keep=relu_keep, pre_out=pre_out)
    i = 4 * n_hidden
    return F.linear(h, p[f"_dnn.{i}.weight"], p[f"_dnn.{i}.bias"]).squeeze(-1)


def dcnv2_forward(x, p: Params, emb: torch.Tensor, num_layers: int, n_hidden: int, training: bool, structure: str):
    """DCNv2.forward after the embedding lookup — src/models/dcn.py:205-222."""
    rows = x + p["offsets"]
    e = emb.reshape(x.shape[0], -1)
    cross = dcn_head(e, p, num_layers, "cross_head.layers")
    if structure == "Stacked":
        logit = bn_mlp(cross, p, "_dnn", n_hidden, training)
    else:
        logit = torch.cat([cross, bn_mlp(e, p, "_dnn", n_hidden, training)], dim=1)
    lin = F.embedding_bag(rows, p["linear_model.weight"], mode="sum")
    return (F.linear(logit, p["_last_fc.weight"], p["_last_fc.bias"]) + lin).squeeze(-1)


# --------------------------------------------------------------------------- TT-Rec (torch semantics)
def tt_forward(indices: torch.Tensor, tt_p_shapes, tt_q_shapes, tt_ranks, cores) -> torch.Tensor:
    """tt_rec_torch_forward — src/models/embeddings/tensortrain_embeddings.py:100-150.  cores[i] is
    [1, p_i, r_i*q_i*r_{i+1}] (tt_ranks has ncores+1 entries); returns [n, prod(q)]."""
    n = len(tt_p_shapes)
    views = [cores[i].view(tt_p_shapes[i], tt_ranks[i], tt_q_shapes[i], tt_ranks[i + 1]).permute(1, 0, 2, 3)
             for i in range(n)]                                    # [r_i, p_i, q_i, r_{i+1}]
    big = 1
    for p in tt_p_shapes:
        big *= int(p)
    res = None
    for i, dim in enumerate(tt_p_shapes):
        big //= int(dim)
        v = indices // big
        indices = indices % big
        sl = views[i][:, v]                                        # [r_i, b, q_i, r_{i+1}]
        if i == 0:
            res = sl
        else:
            res = torch.einsum("abhj,jbkr->abhkr", res, sl)
            res = res.reshape(res.shape[0], res.shape[1], res.shape[2] * res.shape[3], res.shape[4])
    return res.squeeze(0).squeeze(-1)


def masked_topk(user_embs: torch.Tensor, item_embs: torch.Tensor, users: torch.Tensor, graph, k: int,
                filter_item_on_train: bool = True) -> torch.Tensor:
    """Scoring tail of validate_epoch — src/trainer/lightgcn.py:122-138: scores, -inf on the user's train
    items (the reference's Python double loop), torch.topk indices."""
    scores = user_embs[users] @ item_embs.T
    if filter_item_on_train:
        ind0: List[int] = []
        ind1: List[int] = []
        for i, user in enumerate(users.tolist()):
            ind0.extend([i] * len(graph[user]))
            ind1.extend(graph[user])
        scores[ind0, ind1] = float("-inf")
    return torch.topk(scores, k)[1]


# --------------------------------------------------------------------------- QAT (stochastic rounding)
class _StochasticRounding(torch.autograd.Function):
    """StotasticRounding — src/models/embeddings/qat_emb.py:16-84, with the torch.rand_like draw passed in."""

    @staticmethod
    def forward(ctx, scale, w, n_bits: int, prob):
        q_min, q_max = -(1 << (n_bits - 1)), (1 << (n_bits - 1)) - 1
        q_w_float = w / scale
        q_w = torch.clamp(q_w_float, q_min, q_max)
        q_w_floor = torch.floor(q_w)
        prob_floor = q_w_floor + 1 - q_w
        q_w = q_w_floor + (prob > prob_floor)
        ctx.save_for_backward(q_w, q_w_float)
        ctx.q = (q_min, q_max)
        return q_w * scale

    @staticmethod
    def backward(ctx, grad_output):
        res, q_w = ctx.saved_tensors
        q_min, q_max = ctx.q
        scale_mask = torch.empty_like(grad_output)
        mask1, mask2 = q_w >= q_max, q_w <= q_min
        scale_mask[mask1] = q_max
        scale_mask[mask2] = q_min
        mask3 = torch.logical_and(~mask1, ~mask2)
        scale_mask[mask3] = -(q_w[mask3] - res[mask3])
        # autograd reduces the elementwise product to the scalar parameter's shape by summation
        return (grad_output * scale_mask).sum().reshape(()), grad_output.clone(), None, None


def qat_forward(idx: torch.Tensor, weight: torch.Tensor, scale: torch.Tensor, n_bits: int, prob: torch.Tensor):
    """QAT_EmbInt.forward (qat_emb.py:117-119): nn.Embedding rows, then stochastic rounding."""
    return _StochasticRounding.apply(scale, F.embedding(idx, weight), n_bits, prob)


# --------------------------------------------------------------------------- OptEmbed supernet lookup
class _BinaryStep(torch.autograd.Function):
    """optembed_utils.py:25-43."""

    @staticmethod
    def forward(ctx, inp):
        ctx.save_for_backward(inp)
        return (inp > 0.0).float()

    @staticmethod
    def backward(ctx, grad_output):
        (inp,) = ctx.saved_tensors
        additional = 2 - 4 * torch.abs(inp)
        additional[torch.abs(inp) > 1] = 0.0
        additional[(torch.abs(inp) <= 1) * (torch.abs(inp) > 0.4)] = 0.4
        return grad_output.clone() * additional


def optembed_train_forward(x, weight, t_param, mask_d_idx, norm: int = 1):
    """OptEmbed.forward in training (deepfm_opt_embed.py:219-228): x [B,F] rows, one threshold per field, mask_d_idx [B,F]
    the torch.randint draw (number of kept dimensions - 1)."""
    D = weight.shape[1]
    emb = F.embedding(x, weight)
    if t_param is not None:
        emb = emb * _BinaryStep.apply(torch.norm(emb, norm, dim=2) - t_param).unsqueeze(-1)
    full_mask_d = torch.tril(torch.ones((D, D), dtype=torch.bool))
    return F.embedding(mask_d_idx, full_mask_d) * emb


def optembed_weight(weight, t_param, field_dims, mode_threshold_e: str, mask_d_idx=None, mode_threshold_d: str = "field",
                    norm: int = 1):
    """OptEmbed.get_weight in eval (deepfm_opt_embed.py:148-200): the masked table."""
    N, D = weight.shape
    fd = torch.as_tensor(field_dims)
    emb = weight
    if t_param is not None:
        t = t_param if mode_threshold_e == "feature" else torch.repeat_interleave(t_param, fd, dim=0, output_size=N)
        emb = weight * _BinaryStep.apply(torch.norm(weight, norm, dim=1) - t).unsqueeze(-1)
    if mask_d_idx is not None:
        m = mask_d_idx if mode_threshold_d == "feature" else torch.repeat_interleave(mask_d_idx, fd, dim=0, output_size=N)
        emb = emb * F.embedding(m, torch.tril(torch.ones((D, D), dtype=torch.bool))).to(weight)
    return emb


def ndcg_recall(y_pred, y_true, k: int = 20):
    """src/metrics.py:70-108 (get_ndcg_recall; get_ndcg :9-43 is its first component): per user, hits among the first k
    recommendations weighted 1/log2(rank+1) over the ideal DCG of min(len(true), k) hits; recall over the same count."""
    import math

    ndcg = recall = 0.0
    for pred_user, true_user in zip(y_pred, y_true):
        true_user = set(int(t) for t in true_user)
        hits = [1.0 if int(p) in true_user else 0.0 for p in list(pred_user)[:k]]
        dcg = sum(h / math.log2(r + 2) for r, h in enumerate(hits))
        length = min(len(true_user), k)
        idcg = sum(1.0 / math.log2(r + 2) for r in range(length))
        ndcg += dcg / idcg
        recall += sum(hits) / length
    return ndcg / len(y_pred), recall / len(y_pred)
