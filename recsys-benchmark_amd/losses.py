"""Losses around the path.  `BCEWithLogitsLoss` is torch.nn.BCEWithLogitsLoss
(reduction="mean", the reference trainer's criterion, src/trainer/deepfm.py:32,51) as ONE HIP launch
each way instead of ~8 elementwise/reduction launches — at B=4096 every launch is microseconds.
`bpr_loss` / `bpr_loss_rows` are src/losses.py:6-22 (optionally fused with the trainer's three index_selects)
as one HIP launch each way."""
from typing import Optional

import torch
from torch import nn

from . import _kernels, _lib


_unit: dict = {}


def unit_scalar(device) -> torch.Tensor:
    """A resident, read-only fp32 scalar 1 per device: `loss.backward(unit_scalar(dev))` seeds a backward without a fill
    launch, and the fused criterion recognises it (its gradient for an upstream 1 was already written by its forward)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _unit:
        _unit[key] = torch.ones((), dtype=torch.float32, device=torch.device("cuda", key))
    return _unit[key]


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        dev = _lib.require_gpu(logits, target)
        x = _kernels._f32c(logits).view(-1)
        y = _kernels._f32c(target).view(-1)
        if x.numel() != y.numel():
            raise ValueError("logits and target must have the same number of elements")
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        # the gradient for an upstream gradient of 1 comes out of the same launch (16 KB at B = 4096)
        dx_unit = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.load().mi_bce_logits_fwd(x.data_ptr(), y.data_ptr(), loss.data_ptr(), _lib.ptr(dx_unit), x.numel(),
                                                 _lib.stream_ptr(dev)), "mi_bce_logits_fwd")
        ctx.save_for_backward(x, y, *([dx_unit] if dx_unit is not None else []))
        ctx.shape = tuple(logits.shape)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        x, y, *rest = ctx.saved_tensors
        if rest and g.data_ptr() == unit_scalar(x.device).data_ptr():
            return rest[0].view(ctx.shape), None          # seeded with the resident 1: nothing to launch
        g = _kernels._f32c(g).view(1)
        dx = torch.empty_like(x)
        _lib.check(_lib.load().mi_bce_logits_bwd(x.data_ptr(), y.data_ptr(), g.data_ptr(), dx.data_ptr(), x.numel(),
                                                 _lib.stream_ptr(x.device)), "mi_bce_logits_bwd")
        return dx.view(ctx.shape), None


class _HeadBCEFn(torch.autograd.Function):
    """The criterion as the model's head launch already evaluated it (tail.mi_tail_head_bce: DeepFM.forward(x, labels=y)):
    the loss and the gradient for the upstream seed (the resident 1, or the scalar the caller named) exist, nothing is
    launched either way.  Any other upstream gradient takes
    mi_bce_logits_bwd like _BCEFn (and the model's backward then runs its own head backward)."""

    @staticmethod
    def forward(ctx, logits, head):
        ctx.head = head
        ctx.shape = tuple(logits.shape)
        ctx.save_for_backward(logits)
        return head.loss.view(())

    @staticmethod
    def backward(ctx, g):
        head = ctx.head
        (x,) = ctx.saved_tensors
        if g.data_ptr() == head.seed.data_ptr():       # the very scalar the head launch scaled its gradient by
            return head.gvec.view(ctx.shape), None
        g = _kernels._f32c(g).view(1)
        x = x.reshape(-1)
        dx = torch.empty_like(x)
        _lib.check(_lib.load().mi_bce_logits_bwd(x.data_ptr(), head.y.data_ptr(), g.data_ptr(), dx.data_ptr(), x.numel(),
                                                 _lib.stream_ptr(x.device)), "mi_bce_logits_bwd")
        return dx.view(ctx.shape), None


class BCEWithLogitsLoss(nn.Module):
    def forward(self, logits, target):
        from . import tail as _tail

        head = _tail.take_head_loss(logits, target)
        if head is not None:
            return _HeadBCEFn.apply(logits, head)
        return _BCEFn.apply(logits, target.float())


_workspaces = {}
ARMED_WORKSPACES = True


def _armed_workspace(kind: str, dev, elems: int) -> torch.Tensor:
    """A per-(kernel, device, size) workspace whose arrival ticket (the last word) is zero: zeroed here once; the kernels
    that use it leave the ticket zero, so the launch needs no memset node (mi_*_fwd_armed).  Launches that share one must
    run one after the other: fine for everything on one stream (eager steps, one captured graph, its replays — a capture
    runs on torch's capture stream, the warm-up that created the workspace on another, so the key cannot hold the
    stream); a program that runs two such losses of one kind CONCURRENTLY on two streams should call the plain entry
    points (set losses.ARMED_WORKSPACES = False)."""
    if not ARMED_WORKSPACES:
        return None
    key = (kind, str(dev), int(elems))
    ws = _workspaces.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None          # memory allocated inside a capture belongs to that graph's pool: not a thing to keep
        ws = _workspaces[key] = torch.zeros(int(elems), dtype=torch.float32, device=dev)
    return ws


def _ticket_workspace(kind: str, dev, elems: int):
    """(workspace, armed): the kept, pre-zeroed workspace and True, or a fresh one and False (the launch then zeroes the
    ticket with a memset node of its own)."""
    ws = _armed_workspace(kind, dev, elems)
    if ws is not None:
        return ws, True
    return torch.empty(int(elems), dtype=torch.float32, device=dev), False


class _BPRFn(torch.autograd.Function):
    """-logsigmoid(u.p - u.n).mean() with u = U[ui], p = P[pi], n = Nn[ni] (None index = row b)."""

    @staticmethod
    def forward(ctx, U, P, Nn, ui, pi, ni, plus=None, plus_weight: float = 1.0):
        """plus (0-dim tensor, optional): loss = BPR + plus_weight * plus — another term of the objective joined in the
        kernel's last workgroup instead of by a scale launch and an add launch."""
        dev = _lib.require_gpu(U, P, Nn)
        lib = _lib.load()
        U, P, Nn = _kernels._f32c(U), _kernels._f32c(P), _kernels._f32c(Nn)
        idx = [None if t is None else _kernels._i64c(t).view(-1) for t in (ui, pi, ni)]
        B = idx[0].numel() if idx[0] is not None else U.shape[0]
        D = U.shape[1]
        for t, i in zip((U, P, Nn), idx):
            if t.dim() != 2 or t.shape[1] != D or (i.numel() if i is not None else t.shape[0]) != B:
                raise ValueError("bpr_loss: user / positive / negative rows must be [B, D] (or tables with [B] indices)")
        if B == 0:
            raise ValueError("bpr_loss of an empty batch")
        sig = torch.empty(B, dtype=torch.float32, device=dev)
        ws, armed = _ticket_workspace("bpr", dev, lib.mi_bpr_workspace_elems(B))
        loss = torch.empty(2 if plus is not None else 1, dtype=torch.float32, device=dev)
        if plus is not None:
            plus = _kernels._f32c(plus).view(1)
        _lib.check(lib.mi_bpr_fwd_plus(U.data_ptr(), _lib.ptr(idx[0]), P.data_ptr(), _lib.ptr(idx[1]), Nn.data_ptr(),
                                       _lib.ptr(idx[2]), B, D, U.shape[0], P.shape[0], Nn.shape[0],
                                       _lib.err_word(dev).data_ptr(), sig.data_ptr(), ws.data_ptr(), int(armed),
                                       _lib.ptr(plus), float(plus_weight), loss.data_ptr(), _lib.stream_ptr(dev)), "mi_bpr_fwd_plus")
        ctx.save_for_backward(U, P, Nn, sig, *[i for i in idx if i is not None])
        ctx.has_idx = [i is not None for i in idx]
        ctx.meta = (B, D)
        ctx.plus_weight = float(plus_weight) if plus is not None else None
        if plus is not None:          # (sum, bare BPR term): the second is for logging only
            bare = loss[1]
            ctx.mark_non_differentiable(bare)
            ctx.set_materialize_grads(False)      # (no zero-filled "gradient" of the bare term: a fill launch per step)
            return loss[0], bare
        return loss.view(())

    @staticmethod
    def backward(ctx, g, _g_bare=None):
        if g is None:
            return (None,) * 8
        U, P, Nn, sig, *rest = ctx.saved_tensors
        it = iter(rest)
        idx = [next(it) if h else None for h in ctx.has_idx]
        B, D = ctx.meta
        g = _kernels._f32c(g).view(1)
        # positives and negatives picked from ONE table (bpr_loss_rows): one zero-filled gradient takes both contributions
        # — a second one would cost a fill and the add autograd then makes of the two
        need = ctx.needs_input_grad
        same = (idx[1] is not None and idx[2] is not None and need[1] and need[2]
                and P.data_ptr() == Nn.data_ptr() and P.shape == Nn.shape)
        grads = []
        # user and item tables that are the two row segments of ONE matrix (LightGCN's propagation returns them that way):
        # their zero-filled gradients are two views of one buffer — one fill launch instead of two
        joint = None
        if (need[0] and need[1] and idx[0] is not None and idx[1] is not None and U.dim() == 2 and U.shape[1:] == P.shape[1:]
                and U.is_contiguous() and P.is_contiguous() and P.data_ptr() == U.data_ptr() + U.numel() * 4
                and U.untyped_storage().data_ptr() == P.untyped_storage().data_ptr()):
            joint = torch.zeros((U.shape[0] + P.shape[0],) + tuple(U.shape[1:]), dtype=torch.float32, device=U.device)
        for k, (t, i) in enumerate(zip((U, P, Nn), idx)):
            if not need[k]:
                grads.append(None)
            elif k == 2 and same:
                grads.append(grads[1])
            elif joint is not None and k < 2:
                grads.append(joint[: U.shape[0]] if k == 0 else joint[U.shape[0]:])
            else:   # rows repeat under an index array: the kernel accumulates with atomics into zeros
                grads.append(torch.zeros_like(t) if i is not None else torch.empty_like(t))
        _lib.check(_lib.load().mi_bpr_bwd(U.data_ptr(), _lib.ptr(idx[0]), P.data_ptr(), _lib.ptr(idx[1]), Nn.data_ptr(),
                                          _lib.ptr(idx[2]), B, D, U.shape[0], P.shape[0], Nn.shape[0],
                                          sig.data_ptr(), g.data_ptr(), _lib.ptr(grads[0]),
                                          _lib.ptr(grads[1]), _lib.ptr(grads[2]), _lib.stream_ptr(U.device)),
                   "mi_bpr_bwd")
        gplus = None
        if ctx.plus_weight is not None and need[6]:
            # d loss / d plus = plus_weight * g: for the resident unit seed that is a constant, kept per (device, weight)
            if g.data_ptr() == unit_scalar(g.device).data_ptr():
                key = (str(g.device), ctx.plus_weight)
                gplus = _weights.get(key)
                if gplus is None:
                    if torch.cuda.is_current_stream_capturing():
                        gplus = torch.full((), ctx.plus_weight, dtype=torch.float32, device=g.device)
                    else:
                        gplus = _weights[key] = torch.full((), ctx.plus_weight, dtype=torch.float32, device=g.device)
            else:
                gplus = (g * ctx.plus_weight).view(())
        return grads[0], grads[1], (None if same else grads[2]), None, None, None, gplus, None


_weights: dict = {}


def bpr_loss(user_embs, pos_embs, neg_embs):
    """src/losses.py:6-22 — one launch each way."""
    return _BPRFn.apply(user_embs, pos_embs, neg_embs, None, None, None)


def bpr_loss_rows(all_user_emb, all_item_emb, users, pos_items, neg_items, plus=None, plus_weight: float = 1.0,
                  return_parts: bool = False):
    """bpr_loss(index_select(all_user_emb, users), index_select(all_item_emb, pos), index_select(all_item_emb, neg))
    (src/trainer/lightgcn.py:395-399) without materialising the three gathered matrices; the gradients land in
    dense table gradients like index_select's backward.
    plus / plus_weight (optional): returns BPR + plus_weight * plus — the trainer's `loss = loss + reg_weight * reg_loss`
    (src/trainer/lightgcn.py:401-404) inside the same launch; return_parts: (that sum, the bare BPR term for the log)."""
    if plus is None:
        return _BPRFn.apply(all_user_emb, all_item_emb, all_item_emb, users, pos_items, neg_items)
    total, bare = _BPRFn.apply(all_user_emb, all_item_emb, all_item_emb, users, pos_items, neg_items, plus, plus_weight)
    return (total, bare) if return_parts else total


class _RowSqFn(torch.autograd.Function):
    """(|U[ui]|^2 + |P[pi]|^2 + |Nn[ni]|^2) / (2B) over plain tables — LightGCN.get_reg_loss, src/models/lightgcn.py:90-100."""

    @staticmethod
    def forward(ctx, U, P, Nn, ui, pi, ni):
        dev = _lib.require_gpu(U, P, Nn, ui)
        lib = _lib.load()
        U, P, Nn = _kernels._f32c(U), _kernels._f32c(P), _kernels._f32c(Nn)
        ui, pi, ni = (_kernels._i64c(t).view(-1) for t in (ui, pi, ni))
        B, D = ui.numel(), U.shape[1]
        if pi.numel() != B or ni.numel() != B or P.shape[1] != D or Nn.shape[1] != D or B == 0:
            raise ValueError("reg loss: users / positives / negatives must be [B] indices into [*, D] tables")
        ws, armed = _ticket_workspace("rowsq", dev, lib.mi_bpr_workspace_elems(B))
        out = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check((lib.mi_rowsq_fwd_armed if armed else lib.mi_rowsq_fwd)(U.data_ptr(), ui.data_ptr(), P.data_ptr(), pi.data_ptr(), Nn.data_ptr(), ni.data_ptr(),
                                    B, D, U.shape[0], P.shape[0], Nn.shape[0], _lib.err_word(dev).data_ptr(),
                                    ws.data_ptr(), out.data_ptr(), _lib.stream_ptr(dev)), "mi_rowsq_fwd")
        ctx.save_for_backward(U, P, Nn, ui, pi, ni)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        U, P, Nn, ui, pi, ni = ctx.saved_tensors
        g = _kernels._f32c(g).view(1)
        same = P.data_ptr() == Nn.data_ptr() and P.shape == Nn.shape      # positives and negatives share the item table
        dU = torch.zeros_like(U) if ctx.needs_input_grad[0] else None
        dP = torch.zeros_like(P) if (ctx.needs_input_grad[1] or (same and ctx.needs_input_grad[2])) else None
        dN = dP if same else (torch.zeros_like(Nn) if ctx.needs_input_grad[2] else None)
        _lib.check(_lib.load().mi_rowsq_bwd(U.data_ptr(), ui.data_ptr(), P.data_ptr(), pi.data_ptr(), Nn.data_ptr(),
                                            ni.data_ptr(), ui.numel(), U.shape[1], U.shape[0], P.shape[0], Nn.shape[0],
                                            g.data_ptr(), _lib.ptr(dU),
                                            _lib.ptr(dP), _lib.ptr(dN), _lib.stream_ptr(g.device)), "mi_rowsq_bwd")
        # one buffer holds both item contributions when the tables coincide: hand it to the first, nothing to the second
        return dU, dP, (None if same else dN), None, None, None


def reg_loss_rows(user_table, item_table, users, pos_items, neg_items):
    """LightGCN.get_reg_loss on plain [N, D] tables in one launch each way."""
    return _RowSqFn.apply(user_table, item_table, item_table, users, pos_items, neg_items)


def bpr_loss_multi(user_embs, pos_embs, neg_embs):
    """src/losses.py:50-68 — K negatives per positive ([N, K, D]): the BPR kernel over the N*K (user, positive, negative)
    triples, summed over the negatives and averaged over the N samples."""
    if neg_embs.dim() != 3 or neg_embs.shape[0] != user_embs.shape[0]:
        raise ValueError("bpr_loss_multi: negatives must be [N, K, D]")
    n, k = neg_embs.shape[:2]
    rep = torch.arange(n, device=user_embs.device).repeat_interleave(k)
    return _BPRFn.apply(user_embs, pos_embs, neg_embs.reshape(n * k, -1), rep, rep, None) * k


_NORMALIZE_EPS = 1e-12      # F.normalize's default


def _unit_rows(x):
    y, inv = torch.empty_like(x), torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mi_rownorm_fwd(x.data_ptr(), x.shape[0], x.shape[1], _NORMALIZE_EPS, y.data_ptr(), inv.data_ptr(),
                                          _lib.stream_ptr(x.device)), "mi_rownorm_fwd")
    return y, inv


class _InfoNCEFn(torch.autograd.Function):
    """-mean_i log_softmax(v1 v2^T / T)[i, i], rows optionally scaled to unit length first.  The n x n score matrix is
    kept for the backward, which overwrites it with the gradient of the scores (single use)."""

    @staticmethod
    def forward(ctx, v1, v2, temperature, b_cos, valid=None):
        dev = _lib.require_gpu(v1, v2)
        lib = _lib.load()
        if v1.dim() != 2 or v1.shape != v2.shape or v1.shape[0] == 0:
            raise ValueError("info_nce: the two views must be non-empty [N, D] matrices of the same shape")
        same = v1.data_ptr() == v2.data_ptr() and v1.stride() == v2.stride()
        v1 = _kernels._f32c(v1)
        v2 = v1 if same else _kernels._f32c(v2)
        n, D = v1.shape
        inv1 = inv2 = None
        if b_cos:
            v1, inv1 = _unit_rows(v1)
            v2, inv2 = (v1, inv1) if same else _unit_rows(v2)
        S = torch.empty(n, n, dtype=torch.float32, device=dev)
        _kernels.gemm(v1, v2, S, n, n, D, D, D, n, transB=True)
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.mi_lse_diag_workspace_elems(n)), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        inv_t = 1.0 / float(temperature)
        count = None
        if valid is not None:
            if valid.numel() != n:
                raise ValueError("info_nce: `valid` must flag each of the N rows")
            valid = valid.reshape(-1).to(device=dev, dtype=torch.uint8).contiguous()
            count = valid.sum(dtype=torch.float32).view(1)
        _lib.check(lib.mi_lse_diag_fwd(S.data_ptr(), n, n, inv_t, _lib.ptr(valid), _lib.ptr(count), lse.data_ptr(),
                                       ws.data_ptr(), loss.data_ptr(), _lib.stream_ptr(dev)), "mi_lse_diag_fwd")
        ctx.save_for_backward(v1, v2, S, lse, *([inv1, inv2] if b_cos else []))
        ctx.mask = (valid, count)
        ctx.meta = (n, D, inv_t, bool(b_cos))
        ctx.spent = False
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        if ctx.spent:
            raise RuntimeError("info_nce: the saved score matrix was consumed by an earlier backward")
        ctx.spent = True
        v1, v2, S, lse, *invs = ctx.saved_tensors
        n, D, inv_t, b_cos = ctx.meta
        lib = _lib.load()
        dev = S.device
        g = _kernels._f32c(g).view(1)
        valid, count = ctx.mask
        same = v1.data_ptr() == v2.data_ptr()          # one matrix for both views: dS + dS^T formed in place, one product
        _lib.check(lib.mi_lse_diag_bwd(S.data_ptr(), n, n, inv_t, _lib.ptr(valid), _lib.ptr(count), lse.data_ptr(),
                                       g.data_ptr(), int(same), _lib.stream_ptr(dev)), "mi_lse_diag_bwd")
        grads = [None, None]
        for k, (need, other, trans) in enumerate(((ctx.needs_input_grad[0], v2, False), (ctx.needs_input_grad[1], v1, True))):
            if not need or (same and k == 1):
                continue
            d = torch.empty(n, D, dtype=torch.float32, device=dev)
            _kernels.gemm(S, other, d, n, D, n, n, D, D, transA=trans)         # dS v2   /   dS^T v1
            if b_cos:
                y = (v1, v2)[k]
                dx = torch.empty_like(d)
                _lib.check(lib.mi_rownorm_bwd(y.data_ptr(), invs[k].data_ptr(), d.data_ptr(), n, D, _NORMALIZE_EPS,
                                              dx.data_ptr(), _lib.stream_ptr(dev)), "mi_rownorm_bwd")
                d = dx
            grads[k] = d
        if same and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]:
            raise RuntimeError("info_nce(v, v): the gradient is delivered through the first argument")
        return grads[0], grads[1], None, None, None


def info_nce(view1: torch.Tensor, view2: torch.Tensor, temperature: float = 1, b_cos: bool = True,
             valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src/losses.py:25-47.  Row normalisation, the score GEMM (mi_gemm_f32), and one kernel for log-softmax + diagonal +
    mean; the trainer passes the same matrix twice (src/trainer/lightgcn.py:227), which is normalised once.
    valid (extension, [N] bool): rows flagged False are treated as absent — info_nce(v, v, valid=m) == info_nce(v[m], v[m])
    without the data-dependent shape (see first_occurrence)."""
    return _InfoNCEFn.apply(view1, view2, temperature, b_cos, valid)


def first_occurrence(ids: torch.Tensor, num_ids: int) -> torch.Tensor:
    """[B] bool: True where ids[i] appears for the first time in the batch — a fixed-shape stand-in for torch.unique
    (selecting these entries yields each distinct id once).  Two launches, no host sync."""
    ids = ids.reshape(-1)
    pos = torch.arange(ids.numel(), device=ids.device)
    owner = torch.full((num_ids,), ids.numel(), dtype=torch.int64, device=ids.device)
    owner.scatter_reduce_(0, ids, pos, reduce="amin", include_self=True)
    return owner[ids] == pos
