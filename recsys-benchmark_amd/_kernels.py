"""torch.autograd glue over the C-ABI (include/mi355x_recsys.h).

PyTorch is plumbing here: it owns the device buffers and the stream; every
arithmetic step of the hot path runs in libmi355x_recsys.so.
"""
from typing import Optional, Tuple

import torch

from . import _lib


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def _i64c(t: torch.Tensor) -> torch.Tensor:
    # the reference's datasets hand over int32 or int64 ids and `x + offsets`
    # promotes to int64 (src/models/deepfm.py:88); same here.
    if t.dtype not in (torch.int64, torch.int32):
        raise TypeError(f"expected an integer index tensor, got {t.dtype}")
    return t.to(torch.int64).contiguous()


def _coo(rows: torch.Tensor, vals: torch.Tensor, shape) -> torch.Tensor:
    return torch.sparse_coo_tensor(rows.view(1, -1), vals, shape, check_invariants=False)


class GatherFM(torch.autograd.Function):
    """emb, y_fm = gather+FM+first-order (src/models/deepfm.py:88-98) in one kernel.

    sparse_W / sparse_w1 choose the gradient form of the two tables: row form
    (uncoalesced COO, what nn.Embedding(sparse=True) yields) or the reference's
    default dense weight.grad.
    """

    @staticmethod
    def forward(ctx, idx, offsets, W, w1, bias, sparse_W: bool, sparse_w1: bool):
        dev = _lib.require_gpu(idx, offsets, W, w1, bias)
        lib = _lib.load()
        idx = _i64c(idx)
        offsets = _i64c(offsets.reshape(-1))
        Wc, w1c = _f32c(W), _f32c(w1)
        if idx.dim() != 2 or idx.shape[1] != offsets.numel():
            raise ValueError(f"idx must be [B, {offsets.numel()}], got {tuple(idx.shape)}")
        B, F = idx.shape
        N, D = Wc.shape
        if w1c.numel() != N:
            raise ValueError("first-order table must have one weight per embedding row")
        emb = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        yfm = torch.empty((B,), dtype=torch.float32, device=dev)
        rows = torch.empty((B, F), dtype=torch.int64, device=dev)
        _lib.check(
            lib.mi_gather_fm_fwd(
                idx.data_ptr(), offsets.data_ptr(), Wc.data_ptr(), w1c.data_ptr(), _lib.ptr(bias),
                emb.data_ptr(), yfm.data_ptr(), rows.data_ptr(), B, F, D, N,
                _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev),
            ),
            "mi_gather_fm_fwd",
        )
        ctx.save_for_backward(emb, rows)
        ctx.shapes = (B, F, D, N, tuple(W.shape), tuple(w1.shape))
        ctx.sparse = (sparse_W, sparse_w1)
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(rows)
        return emb, yfm, rows

    @staticmethod
    def backward(ctx, g_emb, g_y, _g_rows):
        emb, rows = ctx.saved_tensors
        B, F, D, N, Wshape, w1shape = ctx.shapes
        sparse_W, sparse_w1 = ctx.sparse
        dev = emb.device
        lib = _lib.load()
        if g_y is None:
            g_y = torch.zeros((B,), dtype=torch.float32, device=dev)
        g_y = _f32c(g_y)
        g_emb = None if g_emb is None else _f32c(g_emb)
        need_W, need_w1 = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gW = gw1 = None
        stream = _lib.stream_ptr(dev)
        rows_form = (sparse_W and need_W) or (sparse_w1 and need_w1)
        dense_form = (need_W and not sparse_W) or (need_w1 and not sparse_w1)
        if rows_form:
            gvals = torch.empty((B * F, D), dtype=torch.float32, device=dev)
            g1vals = torch.empty((B * F,), dtype=torch.float32, device=dev)
            _lib.check(
                lib.mi_gather_fm_bwd_rows(emb.data_ptr(), g_y.data_ptr(), _lib.ptr(g_emb),
                                          gvals.data_ptr(), g1vals.data_ptr(), B, F, D, stream),
                "mi_gather_fm_bwd_rows",
            )
            if sparse_W and need_W:
                gW = _coo(rows, gvals, Wshape)
            if sparse_w1 and need_w1:
                gw1 = _coo(rows, g1vals.view((-1,) + (1,) * (len(w1shape) - 1)), w1shape)
        if dense_form:
            gWd = torch.zeros((N, D), dtype=torch.float32, device=dev)
            gw1d = torch.zeros((N,), dtype=torch.float32, device=dev)
            _lib.check(
                lib.mi_gather_fm_bwd_dense(rows.data_ptr(), emb.data_ptr(), g_y.data_ptr(),
                                           _lib.ptr(g_emb), gWd.data_ptr(), gw1d.data_ptr(),
                                           B, F, D, N, stream),
                "mi_gather_fm_bwd_dense",
            )
            if need_W and not sparse_W:
                gW = gWd.view(Wshape)
            if need_w1 and not sparse_w1:
                gw1 = gw1d.view(w1shape)
        gb = g_y.sum().view(1) if (ctx.has_bias and ctx.needs_input_grad[4]) else None
        return None, None, gW, gw1, gb, None, None


def gather_fm(idx, offsets, W, w1, bias, sparse_W=False, sparse_w1=False):
    emb, yfm, _rows = GatherFM.apply(idx, offsets, W, w1, bias, sparse_W, sparse_w1)
    return emb, yfm


class GatherRows(torch.autograd.Function):
    """out[i,:] = W[idx[i],:]  (nn.Embedding, src/models/embeddings/base.py:74-75)."""

    @staticmethod
    def forward(ctx, idx, W, sparse: bool):
        dev = _lib.require_gpu(idx, W)
        lib = _lib.load()
        idxc = _i64c(idx)
        Wc = _f32c(W)
        N, D = Wc.shape
        n = idxc.numel()
        out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=dev)
        _lib.check(
            lib.mi_gather_rows_fwd(idxc.data_ptr(), Wc.data_ptr(), out.data_ptr(), n, D, N,
                                   _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
            "mi_gather_rows_fwd",
        )
        ctx.save_for_backward(idxc)
        ctx.meta = (n, D, N, tuple(W.shape), sparse)
        return out

    @staticmethod
    def backward(ctx, g):
        (idxc,) = ctx.saved_tensors
        n, D, N, Wshape, sparse = ctx.meta
        if not ctx.needs_input_grad[1]:
            return None, None, None
        g = _f32c(g).view(n, D)
        if sparse:
            return None, _coo(idxc.view(-1), g, Wshape), None
        gW = torch.zeros((N, D), dtype=torch.float32, device=g.device)
        _lib.check(
            _lib.load().mi_scatter_add_rows(idxc.data_ptr(), g.data_ptr(), gW.data_ptr(), n, D, N,
                                            _lib.stream_ptr(g.device)),
            "mi_scatter_add_rows",
        )
        return None, gW.view(Wshape), None


def gather_rows(idx: torch.Tensor, W: torch.Tensor, sparse: bool = False) -> torch.Tensor:
    return GatherRows.apply(idx, W, sparse)


def bag_reduce(rows: torch.Tensor, mode: Optional[str]) -> torch.Tensor:
    """EmbeddingBag semantics for a 2-D index tensor: every row of the input is a bag
    (src/models/embeddings/base.py:58-63).  rows: [B, L, D]."""
    if mode is None:
        return rows
    if rows.dim() != 3:
        raise ValueError("bag modes need a 2-D index tensor (one bag per row)")
    if mode == "sum":
        return rows.sum(1)
    if mode == "mean":
        return rows.mean(1)
    if mode == "max":
        return rows.max(1)[0]
    raise ValueError(f"unknown mode {mode}")
