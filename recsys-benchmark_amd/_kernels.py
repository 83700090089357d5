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


# --------------------------------------------------------------------------------------
# FM + first-order over an already gathered emb (DeepFM on a compressed table)
class FMFirstOrder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, rows, w1, bias, sparse_w1: bool):
        dev = _lib.require_gpu(emb, rows, w1, bias)
        embc, rowsc, w1c = _f32c(emb), _i64c(rows), _f32c(w1)
        B, F, D = embc.shape
        N = w1c.numel()
        yfm = torch.empty((B,), dtype=torch.float32, device=dev)
        _lib.check(
            _lib.load().mi_fm_fwd(embc.data_ptr(), rowsc.data_ptr(), w1c.data_ptr(), _lib.ptr(bias),
                                  yfm.data_ptr(), B, F, D, N, _lib.err_word(dev).data_ptr(),
                                  _lib.stream_ptr(dev)),
            "mi_fm_fwd",
        )
        ctx.save_for_backward(embc, rowsc)
        ctx.meta = (B, F, D, N, tuple(w1.shape), sparse_w1, bias is not None)
        return yfm

    @staticmethod
    def backward(ctx, g_y):
        embc, rowsc = ctx.saved_tensors
        B, F, D, N, w1shape, sparse_w1, has_bias = ctx.meta
        dev = embc.device
        lib = _lib.load()
        g_y = _f32c(g_y)
        stream = _lib.stream_ptr(dev)
        g_emb = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        g1vals = torch.empty((B * F,), dtype=torch.float32, device=dev)
        _lib.check(
            lib.mi_gather_fm_bwd_rows(embc.data_ptr(), g_y.data_ptr(), None, g_emb.data_ptr(),
                                      g1vals.data_ptr(), B, F, D, stream),
            "mi_gather_fm_bwd_rows",
        )
        gw1 = None
        if ctx.needs_input_grad[2]:
            if sparse_w1:
                gw1 = _coo(rowsc, g1vals.view((-1,) + (1,) * (len(w1shape) - 1)), w1shape)
            else:
                gw1 = torch.zeros((N,), dtype=torch.float32, device=dev)
                _lib.check(lib.mi_scatter_add_rows(rowsc.data_ptr(), g1vals.data_ptr(), gw1.data_ptr(),
                                                   B * F, 1, N, stream), "mi_scatter_add_rows")
                gw1 = gw1.view(w1shape)
        gb = g_y.sum().view(1) if (has_bias and ctx.needs_input_grad[3]) else None
        return (g_emb if ctx.needs_input_grad[0] else None), None, gw1, gb, None


def fm_first_order(emb, rows, w1, bias, sparse_w1=False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(emb, y_fm) for DeepFM when emb came from any IEmbedding (src/models/deepfm.py:91-98)."""
    if emb.dim() != 3 or tuple(rows.shape) != tuple(emb.shape[:2]):
        # QR 'cat' yields [B,2F,D/2] (reference quirk); the FM terms are defined over that tensor
        # while the first-order bag still runs over the B x F ids.
        return emb, _fm_first_order_mismatched(emb, rows, w1, bias, sparse_w1)
    return emb, FMFirstOrder.apply(emb, rows, w1, bias, sparse_w1)


def _fm_first_order_mismatched(emb, rows, w1, bias, sparse_w1):
    B = emb.shape[0]
    zero_rows = torch.zeros((B, emb.shape[1]), dtype=torch.int64, device=emb.device)
    # second-order part over emb's own [B,F',D'] shape with a zero first-order table ...
    zero_w1 = torch.zeros((1,), dtype=torch.float32, device=emb.device)
    second = FMFirstOrder.apply(emb, zero_rows, zero_w1, None, False)
    # ... plus the first-order bag over the real ids: a [B,F,1] "embedding" whose FM term is
    # 0.5*((sum w)^2 - sum w^2); subtract it back out by computing only the linear part.
    lin = GatherRows.apply(rows, w1.view(-1, 1), sparse_w1).sum(dim=(1, 2))
    return second + lin + (bias if bias is not None else 0.0)


# --------------------------------------------------------------------------------------
OPS = {"mult": 0, "add": 1, "cat": 2}
XF_NONE, XF_SOFT, XF_MASK = 0, 1, 2


class DualGather(torch.autograd.Function):
    """out = T1'[idx % mod1] (op) T2'[idx // div2]  (mi_dual_gather_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, idx, T1, T2, S1, S2, M1, M2, mod1: int, div2: int, op: int, xform: int):
        dev = _lib.require_gpu(idx, T1, T2)
        idxc = _i64c(idx)
        T1c, T2c = _f32c(T1), _f32c(T2)
        S1c = None if S1 is None else _f32c(S1)
        S2c = None if S2 is None else _f32c(S2)
        M1c = None if M1 is None else M1.to(torch.uint8).contiguous()
        M2c = None if M2 is None else M2.to(torch.uint8).contiguous()
        n = idxc.numel()
        De = T1c.shape[1]
        if T2c.shape[1] != De:
            raise ValueError("both tables must share the row width")
        F = idx.shape[1] if idx.dim() == 2 else 1
        if op == OPS["cat"]:
            if idx.dim() == 1:
                oshape = (idx.shape[0], 2 * De)
            elif idx.dim() == 2:
                oshape = (idx.shape[0], 2 * F, De)
            else:
                raise ValueError("cat supports 1-D or 2-D index tensors")
        else:
            oshape = tuple(idx.shape) + (De,)
        out = torch.empty(oshape, dtype=torch.float32, device=dev)
        _lib.check(
            _lib.load().mi_dual_gather_fwd(
                idxc.data_ptr(), T1c.data_ptr(), T2c.data_ptr(), _lib.ptr(S1c), _lib.ptr(S2c),
                _lib.ptr(M1c), _lib.ptr(M2c), out.data_ptr(), n, F, De, T1c.shape[0], T2c.shape[0],
                mod1, div2, op, xform, _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
            "mi_dual_gather_fwd",
        )
        ctx.save_for_backward(idxc, T1c, T2c, S1c, S2c, M1c, M2c)
        ctx.meta = (n, F, De, mod1, div2, op, xform)
        return out

    @staticmethod
    def backward(ctx, g):
        idxc, T1c, T2c, S1c, S2c, M1c, M2c = ctx.saved_tensors
        n, F, De, mod1, div2, op, xform = ctx.meta
        dev = g.device
        g = _f32c(g)
        gT1, gT2 = torch.zeros_like(T1c), torch.zeros_like(T2c)
        gS1 = torch.zeros_like(S1c) if xform == XF_SOFT else None
        gS2 = torch.zeros_like(S2c) if xform == XF_SOFT else None
        _lib.check(
            _lib.load().mi_dual_gather_bwd(
                idxc.data_ptr(), g.data_ptr(), T1c.data_ptr(), T2c.data_ptr(), _lib.ptr(S1c), _lib.ptr(S2c),
                _lib.ptr(M1c), _lib.ptr(M2c), gT1.data_ptr(), gT2.data_ptr(), _lib.ptr(gS1), _lib.ptr(gS2),
                n, F, De, T1c.shape[0], T2c.shape[0], mod1, div2, op, xform, _lib.stream_ptr(dev)),
            "mi_dual_gather_bwd",
        )
        return None, gT1, gT2, gS1, gS2, None, None, None, None, None, None


def dual_gather(idx, T1, T2, mod1, div2, op="add", S1=None, S2=None, M1=None, M2=None):
    xform = XF_SOFT if S1 is not None else (XF_MASK if M1 is not None else XF_NONE)
    return DualGather.apply(idx, T1, T2, S1, S2, M1, M2, int(mod1), int(div2), OPS[op], xform)


def csr_rows(values, crow, col, ids, D: int, N: int) -> torch.Tensor:
    """Dense rows of a CSR-stored table (inference only, like the reference's PrunedEmbedding)."""
    dev = _lib.require_gpu(crow, ids)
    idsc = _i64c(ids)
    out = torch.empty(tuple(ids.shape) + (D,), dtype=torch.float32, device=dev)
    _lib.check(
        _lib.load().mi_csr_rows_fwd(_lib.ptr(values), crow.data_ptr(), _lib.ptr(col), idsc.data_ptr(),
                                    out.data_ptr(), idsc.numel(), D, N, _lib.err_word(dev).data_ptr(),
                                    _lib.stream_ptr(dev)),
        "mi_csr_rows_fwd",
    )
    return out


def dhe_hash(ids, slopes, bias, primes, prefix: int, m: int) -> torch.Tensor:
    """[n] int64 ids -> [n, k] fp32 hash features in [-1, 1] (not differentiable)."""
    dev = _lib.require_gpu(ids, slopes, bias, primes)
    idsc = _i64c(ids)
    K = slopes.numel()
    out = torch.empty(tuple(ids.shape) + (K,), dtype=torch.float32, device=dev)
    _lib.check(
        _lib.load().mi_dhe_hash(idsc.data_ptr(), slopes.contiguous().data_ptr(), bias.contiguous().data_ptr(),
                                primes.contiguous().data_ptr(), out.data_ptr(), idsc.numel(), K, int(prefix),
                                int(m), _lib.stream_ptr(dev)),
        "mi_dhe_hash",
    )
    return out
