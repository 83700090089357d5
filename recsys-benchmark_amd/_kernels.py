"""torch.autograd glue over the C-ABI (include/mi355x_recsys.h).

PyTorch is plumbing here: it owns the device buffers and the stream; every
arithmetic step of the hot path runs in libmi355x_recsys.so.
"""
import collections
import ctypes as _ctypes
from typing import Optional, Tuple

import torch

from . import _lib


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def _float4_rows(D: int) -> bool:
    """Row widths the float4 kernels take (csrc: vec_ok): D = 4 * 2^k <= 256."""
    return D % 4 == 0 and 4 <= D <= 256 and ((D // 4) & (D // 4 - 1)) == 0


def _row_strided(t: torch.Tensor, align: int = 4):
    """(tensor, row stride in floats) of a 2-D fp32 table the kernels can read in place: unit stride inside a row, any
    row stride that is a multiple of `align` floats (4 keeps float4 row accesses 16-byte aligned) — a column-slice view
    of a packed table, DeepFM.pack_tables(); anything else is made contiguous."""
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if (t.dim() == 2 and t.shape[0] > 1 and (t.shape[1] == 1 or t.stride(1) == 1) and t.stride(0) >= t.shape[1]
            and t.stride(0) % align == 0 and (t.data_ptr() % (4 * align)) == 0):
        return t, int(t.stride(0))
    t = t.contiguous()
    return t, int(t.shape[1]) if t.dim() == 2 else 1


def _i64c(t: torch.Tensor) -> torch.Tensor:
    # the reference's datasets hand over int32 or int64 ids and `x + offsets`
    # promotes to int64 (src/models/deepfm.py:88); same here.
    if t.dtype not in (torch.int64, torch.int32):
        raise TypeError(f"expected an integer index tensor, got {t.dtype}")
    return t.to(torch.int64).contiguous()


def _coo(rows: torch.Tensor, vals: torch.Tensor, shape) -> torch.Tensor:
    return torch.sparse_coo_tensor(rows.view(1, -1), vals, shape, check_invariants=False)


# Row ids that came from a multi-field lookup (rows[b, f] = x[b, f] + offsets[f]) and may later head a row-form
# gradient: the optimizer can sort them field by field in LDS (mi_sort_field_rows) instead of generically.  The entry
# holds the ids tensor itself, so its storage — and with it the pointer used as the key — cannot be recycled while the
# entry lives; only the last few lookups are kept.
_field_layouts: "collections.OrderedDict[int, tuple]" = collections.OrderedDict()


def note_field_layout(rows: torch.Tensor, offsets: torch.Tensor, N: int) -> None:
    """rows int64[B, F] = x + offsets (offsets ascending, field f's ids in [offsets[f], offsets[f+1]), the last below N)."""
    if rows.dim() != 2 or not rows.is_contiguous() or rows.dtype != torch.int64 or not rows.is_cuda:
        return
    offsets = offsets.reshape(-1)
    _field_layouts[rows.data_ptr()] = (rows, offsets, N)
    _field_layouts.move_to_end(rows.data_ptr())
    while len(_field_layouts) > 4:
        _field_layouts.popitem(last=False)


def sort_field_rows(rows_flat: torch.Tensor, N: int):
    """(sorted ids, permutation) of a flat view of ids noted by a multi-field lookup, or None when the ids are not known
    to have that layout (or exceed the LDS sort's limits) and the caller has to sort them generically."""
    entry = _field_layouts.get(rows_flat.data_ptr())
    if entry is None:
        return None
    rows, offsets, n_rows = entry
    B, F = rows.shape
    if rows_flat.numel() != B * F or n_rows != N or B > 65536 or N >= 2**32 - 2 or B == 0:
        return None
    lib = _lib.load()
    rows_sorted, perm = torch.empty_like(rows_flat), torch.empty_like(rows_flat)
    nbytes = int(lib.mi_sort_field_rows_workspace_bytes(B, F))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=rows.device) if nbytes else None
    _lib.check(lib.mi_sort_field_rows(rows.data_ptr(), offsets.data_ptr(), N, B, F, rows_sorted.data_ptr(),
                                      perm.data_ptr(), _lib.ptr(ws), _lib.err_word(rows.device).data_ptr(),
                                      _lib.stream_ptr(rows.device)), "mi_sort_field_rows")
    return rows_sorted, perm


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
        if _float4_rows(W.shape[-1]):
            Wc, ldw = _row_strided(W)
            w1c, ldw1 = _row_strided(w1.reshape(w1.shape[0], -1) if w1.dim() != 2 else w1, align=1)
        else:       # widths the float4 kernels do not cover run on the scalar kernels, which read the two plain tensors
            Wc, w1c = _f32c(W), _f32c(w1)
            ldw, ldw1 = int(Wc.shape[-1]), 1
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
            lib.mi_gather_fm_fwd_ld(
                idx.data_ptr(), offsets.data_ptr(), Wc.data_ptr(), ldw, w1c.data_ptr(), ldw1, _lib.ptr(bias),
                emb.data_ptr(), yfm.data_ptr(), rows.data_ptr(), B, F, D, N,
                _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev),
            ),
            "mi_gather_fm_fwd_ld",
        )
        if sparse_W or sparse_w1 or DETERMINISTIC:
            note_field_layout(rows, offsets, N)
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
        if DETERMINISTIC and dense_form:
            rows_form, dense_form = True, False     # row-form values, then a sorted (ordered) accumulation: no atomics
        # the bias gradient (sum of g_y) rides in whichever backward kernel runs first
        gb = torch.empty((1,), dtype=torch.float32, device=dev) if (ctx.has_bias and ctx.needs_input_grad[4]) else None
        gb_done = False
        if rows_form:
            gvals = torch.empty((B * F, D), dtype=torch.float32, device=dev)
            g1vals = torch.empty((B * F,), dtype=torch.float32, device=dev)
            _lib.check(
                lib.mi_gather_fm_bwd_rows(emb.data_ptr(), g_y.data_ptr(), _lib.ptr(g_emb),
                                          gvals.data_ptr(), g1vals.data_ptr(), _lib.ptr(gb), B, F, D, stream),
                "mi_gather_fm_bwd_rows",
            )
            gb_done = True
            # a table that wants a dense gradient next to one in row form (the reference's sparse config keeps the
            # first-order table dense): scatter-add the values just computed — never a second pass over a zeroed [N, D]
            if need_W:
                gW = _coo(rows, gvals, Wshape) if sparse_W else _scatter_rows(rows, gvals, N, D, stream).view(Wshape)
            if need_w1:
                gw1 = (_coo(rows, g1vals.view((-1,) + (1,) * (len(w1shape) - 1)), w1shape) if sparse_w1
                       else _scatter_rows(rows, g1vals, N, 1, stream).view(w1shape))
        elif dense_form:
            gWd = torch.zeros((N, D), dtype=torch.float32, device=dev)
            gw1d = torch.zeros((N,), dtype=torch.float32, device=dev)
            _lib.check(
                lib.mi_gather_fm_bwd_dense(rows.data_ptr(), emb.data_ptr(), g_y.data_ptr(),
                                           _lib.ptr(g_emb), gWd.data_ptr(), gw1d.data_ptr(),
                                           _lib.ptr(gb), B, F, D, N, stream),
                "mi_gather_fm_bwd_dense",
            )
            gb_done = True
            if need_W:
                gW = gWd.view(Wshape)
            if need_w1:
                gw1 = gw1d.view(w1shape)
        if gb is not None and not gb_done:          # neither table needs a gradient: nothing was launched
            gb = g_y.sum().view(1)
        return None, None, gW, gw1, gb, None, None


# Deterministic mode (recsys_benchmark_amd.use_deterministic_algorithms): dense table gradients are built by SORTING the
# row ids and adding each row's contributions in a fixed order (mi_coalesce_rows_sorted) instead of float atomics — the
# reference's CPU index_add is deterministic too (src/models/embeddings/base.py:74-75 through autograd) —, the products
# that would split K over atomically accumulating workgroups run unsplit, and the MLP tail runs on the fused kernels of
# csrc/tail.hip (no atomics).  Off by default: the atomic forms are faster.
DETERMINISTIC = False


def _no_atomics_promised(what: str) -> None:
    """use_deterministic_algorithms(True) promises gradients without float atomics; a backward that only exists in the
    atomic form says so instead of handing out gradients that merely look reproducible."""
    if DETERMINISTIC:
        raise NotImplementedError(f"{what} accumulates duplicate rows with float atomics; it has no deterministic form "
                                  "(recsys_benchmark_amd.use_deterministic_algorithms(True) is on)")


def coalesce_dense(rows: torch.Tensor, vals: torch.Tensor, N: int, D: int) -> torch.Tensor:
    """Dense [N, D] gradient of a table out of row-form (rows[n], vals[n, D]) contributions, added in a fixed order."""
    dev = vals.device
    rows = rows.reshape(-1)
    n = rows.numel()
    out = torch.zeros((N, D), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    found = sort_field_rows(rows, N)
    rs, perm = found if found is not None else torch.sort(rows, stable=True)
    acc = torch.empty((n, D), dtype=torch.float32, device=dev)
    vals = _f32c(vals).view(n, D)
    _lib.check(_lib.load().mi_coalesce_rows_sorted(rs.data_ptr(), perm.data_ptr(), vals.data_ptr(), out.data_ptr(), acc.data_ptr(),
                                                   n, D, N, _lib.stream_ptr(dev)), "mi_coalesce_rows_sorted")
    return out


def _scatter_rows(rows, vals, N, D, stream):
    if DETERMINISTIC:
        return coalesce_dense(rows, vals, N, D)
    out = torch.zeros((N, D), dtype=torch.float32, device=vals.device)
    _lib.check(_lib.load().mi_scatter_axpy_rows(rows.data_ptr(), vals.data_ptr(), 1.0, out.data_ptr(), rows.numel(), D, N,
                                                stream), "mi_scatter_axpy_rows")
    return out


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
        if DETERMINISTIC:
            return None, coalesce_dense(idxc, g, N, D).view(Wshape), None
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
        gb = torch.empty((1,), dtype=torch.float32, device=dev) if (has_bias and ctx.needs_input_grad[3]) else None
        _lib.check(
            lib.mi_gather_fm_bwd_rows(embc.data_ptr(), g_y.data_ptr(), None, g_emb.data_ptr(),
                                      g1vals.data_ptr(), _lib.ptr(gb), B, F, D, stream),
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
        return (g_emb if ctx.needs_input_grad[0] else None), None, gw1, gb, None


def fm_first_order(emb, rows, w1, bias, sparse_w1=False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(emb, y_fm) for DeepFM when emb came from any IEmbedding (src/models/deepfm.py:91-98)."""
    if emb.dim() != 3 or tuple(rows.shape) != tuple(emb.shape[:2]):
        return emb, _fm_first_order_mismatched(emb, rows, w1, bias, sparse_w1)
    return emb, FMFirstOrder.apply(emb, rows, w1, bias, sparse_w1)


def _fm_first_order_mismatched(emb, rows, w1, bias, sparse_w1):
    """QR with operation="cat": emb is [B, 2F, D/2] while the ids are [B, F].  The second-order term runs
    over emb's own shape (mi_fm_fwd with an all-zero first-order table), the first-order bag over the
    real ids (a D=1 row gather summed per sample)."""
    B = emb.shape[0]
    zero_rows = torch.zeros((B, emb.shape[1]), dtype=torch.int64, device=emb.device)
    zero_w1 = torch.zeros((1,), dtype=torch.float32, device=emb.device)
    second = FMFirstOrder.apply(emb, zero_rows, zero_w1, None, False)
    lin = GatherRows.apply(rows, w1.view(-1, 1), sparse_w1).sum(dim=(1, 2))
    return second + lin + (bias if bias is not None else 0.0)


# --------------------------------------------------------------------------------------
OPS = {"mult": 0, "add": 1, "cat": 2}
XF_NONE, XF_SOFT, XF_MASK = 0, 1, 2


class DualGather(torch.autograd.Function):
    """out = T1'[idx % mod1] (op) T2'[idx // div2]  (mi_dual_gather_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, idx, T1, T2, S1, S2, M1, M2, mod1: int, div2: int, op: int, xform: int, fields=None, sparse2: bool = False,
                offsets=None):
        """offsets ([F], optional): idx holds the model's raw per-field ids and the per-field row offsets are added inside
        the lookup; the second output is then the row ids idx + offsets (int64, not differentiable)."""
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
        rows = None
        if offsets is not None:
            if idx.dim() != 2 or offsets.numel() != F:
                raise ValueError("offsets go with [B, F] ids, one per field")
            offs = _i64c(offsets.reshape(-1))
            rows = torch.empty_like(idxc)
        _lib.check(
            _lib.load().mi_dual_gather_fwd_off(
                idxc.data_ptr(), _lib.ptr(offs) if rows is not None else None, _lib.ptr(rows), T1c.data_ptr(), T2c.data_ptr(),
                _lib.ptr(S1c), _lib.ptr(S2c), _lib.ptr(M1c), _lib.ptr(M2c), out.data_ptr(), n, F, De, T1c.shape[0], T2c.shape[0],
                mod1, div2, op, xform, _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
            "mi_dual_gather_fwd_off",
        )
        if rows is not None:
            idxc = rows              # the backward reads the finished row ids
            ctx.mark_non_differentiable(rows)
            # (autograd would otherwise hand the backward a ZERO-FILLED int64 [B, F] "gradient" for the row ids: a fill launch
            #  per step for a tensor nobody reads)
            ctx.set_materialize_grads(False)
        ctx.save_for_backward(idxc, T1c, T2c, S1c, S2c, M1c, M2c)
        ctx.meta = (n, F, De, mod1, div2, op, xform)
        ctx.fields = fields if (fields is not None and idx.dim() == 2 and fields[3] == F) else None
        ctx.sparse2 = bool(sparse2) and xform == XF_NONE
        ctx.t2_shape = tuple(T2.shape)
        ctx.with_rows = rows is not None
        return (out, rows) if rows is not None else out

    @staticmethod
    def backward(ctx, g, _g_rows=None):
        if g is None:                # (materialisation is off when the row ids are a second output: nothing flowed back)
            return (None,) * 14
        idxc, T1c, T2c, S1c, S2c, M1c, M2c = ctx.saved_tensors
        n, F, De, mod1, div2, op, xform = ctx.meta
        dev = g.device
        _no_atomics_promised("the two-table (QR / CERP) gather backward")
        g = _f32c(g)
        if ctx.sparse2:
            # table 2 (the quotient table: ~N / divider rows) gets its gradient in ROW form: one value row per lookup, no
            # atomics into scattered rows and no [n2, De] zero-fill; table 1 (divider rows) stays dense
            ws = _dual_rows_workspace(dev, De, T1c.shape[0])
            # (with the workspace the last workgroup WRITES table 1's gradient: no zero fill)
            written = ws is not None and _lib.load().mi_dual_gather_bwd_rows_overwrites(De, T1c.shape[0]) != 0
            gT1 = torch.empty_like(T1c) if written else torch.zeros_like(T1c)
            g2vals = torch.empty((n, De), dtype=torch.float32, device=dev)
            rows2 = torch.empty((n,), dtype=torch.int64, device=dev)
            _lib.check(_lib.load().mi_dual_gather_bwd_rows(idxc.data_ptr(), g.data_ptr(), T1c.data_ptr(), T2c.data_ptr(), gT1.data_ptr(),
                                                           g2vals.data_ptr(), rows2.data_ptr(), n, F, De, T1c.shape[0], T2c.shape[0],
                                                           mod1, div2, op, _lib.ptr(ws), _lib.stream_ptr(dev)), "mi_dual_gather_bwd_rows")
            # (ids out of range carry row 0 and a zero value row: they add nothing)
            return None, gT1, _coo(rows2, g2vals, ctx.t2_shape), None, None, None, None, None, None, None, None, None, None, None
        gT1, gT2 = torch.zeros_like(T1c), torch.zeros_like(T2c)
        gS1 = torch.zeros_like(S1c) if xform == XF_SOFT else None
        gS2 = torch.zeros_like(S2c) if xform == XF_SOFT else None
        small, row0, flags = (ctx.fields[0], ctx.fields[1], ctx.fields[2]) if ctx.fields is not None else (None, None, None)
        _lib.check(
            _lib.load().mi_dual_gather_bwd_fields(
                idxc.data_ptr(), g.data_ptr(), T1c.data_ptr(), T2c.data_ptr(), _lib.ptr(S1c), _lib.ptr(S2c),
                _lib.ptr(M1c), _lib.ptr(M2c), gT1.data_ptr(), gT2.data_ptr(), _lib.ptr(gS1), _lib.ptr(gS2),
                n, F, De, T1c.shape[0], T2c.shape[0], mod1, div2, op, xform, _lib.ptr(small),
                small.numel() if small is not None else 0, _lib.ptr(row0), _lib.ptr(flags), _lib.stream_ptr(dev)),
            "mi_dual_gather_bwd_fields",
        )
        return None, gT1, gT2, gS1, gS2, None, None, None, None, None, None, None, None, None


_DUAL_WS = {}


def _dual_rows_workspace(dev, De: int, n1: int):
    """The persistent join workspace of mi_dual_gather_bwd_rows for this device and shape (its ticket word is zero between
    launches: allocated zeroed once, the kernel resets it), or None when the shape has no use for one."""
    key = (str(dev), int(De), int(n1))
    if key not in _DUAL_WS:
        ne = int(_lib.load().mi_dual_gather_bwd_rows_workspace_elems(De, n1))
        _DUAL_WS[key] = torch.zeros((ne,), dtype=torch.float32, device=dev) if ne else None
    return _DUAL_WS[key]


SMALL_FIELD_ROWS = 16     # kSmallRows of csrc/embed.hip


def small_field_hint(field_dims, div2: int, device):
    """The `fields` hint of dual_gather for ids that are per-field ids + cumulative offsets (how the CTR models address one
    shared table): (small field indices int32, first table-2 row of every field int64, flags uint8, F), or None when no
    field is small.  A field is small when its ids reach at most 16 rows of table 2 (idx // div2)."""
    if isinstance(field_dims, int) or len(field_dims) == 0:
        return None
    off, row0, small = 0, [], []
    for f, card in enumerate(field_dims):
        lo, hi = off // div2, (off + card - 1) // div2
        row0.append(lo)
        if hi - lo + 1 <= SMALL_FIELD_ROWS:
            small.append(f)
        off += card
    if not small:
        return None
    flags = torch.zeros(len(field_dims), dtype=torch.uint8)
    flags[small] = 1
    return (torch.tensor(small, dtype=torch.int32, device=device), torch.tensor(row0, dtype=torch.int64, device=device),
            flags.to(device), len(field_dims))


def dual_gather(idx, T1, T2, mod1, div2, op="add", S1=None, S2=None, M1=None, M2=None, fields=None, sparse2=False, offsets=None):
    """fields (optional): small_field_hint(...) — lets the backward sum the gradient of low-cardinality fields per field
    instead of with thousands of same-address atomics; same result up to the order of float additions.
    offsets (optional, [F]): idx holds raw per-field ids; returns (out, idx + offsets), the addition done by the lookup."""
    xform = XF_SOFT if S1 is not None else (XF_MASK if M1 is not None else XF_NONE)
    return DualGather.apply(idx, T1, T2, S1, S2, M1, M2, int(mod1), int(div2), OPS[op], xform, fields, bool(sparse2), offsets)


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


# --------------------------------------------------------------------------------------
# LightGCN propagation: CSR SpMM with fused layer sum
HUB_DEGREE = 256  # rows with more nonzeros than this get a whole workgroup


def _os_env(name, default):
    import os
    return os.environ.get(name, default)


class CsrPlan:
    """int32 CSR of A and of A^T plus the short/hub row split, built once per sparsity pattern.

    The reference passes a torch sparse-CSR tensor with int64 indices every call
    (src/trainer/lightgcn.py:36-40); the plan is cached on its index tensors.  Values are read
    fresh on every call (SparseDropout changes them, src/models/layers.py:15-37).
    """

    def __init__(self, crow: torch.Tensor, col: torch.Tensor, shape):
        n_rows, n_cols = shape
        nnz = col.numel()
        if nnz >= 2 ** 31 or max(n_rows, n_cols) >= 2 ** 31:
            raise ValueError("matrix too large for int32 CSR indices")
        self.shape = (n_rows, n_cols)
        self.nnz = nnz
        dev = crow.device
        # the plan is cached under the ADDRESSES of the caller's index tensors (csr_plan): keep them alive so that no
        # other graph's indices can be allocated there while the entry exists
        self.src_crow, self.src_col = crow, col
        self.src_versions = (crow._version, col._version)
        self.crow = crow.to(torch.int32).contiguous()
        self.col = col.to(torch.int32).contiguous()
        deg = (crow[1:] - crow[:-1])
        self.short_rows, self.long_rows = self._split(deg)
        self.is_hub = (deg > HUB_DEGREE).to(torch.uint8)          # per row of A: handled by a whole workgroup
        self.hub_need = torch.zeros(n_rows, dtype=torch.uint8, device=dev)     # mi_spmm_csr_sel's flags: zero between launches
        self.filler_row = int(self.short_rows[0]) if self.short_rows.numel() else -1
        # transpose: sort entries by (col, row); perm maps transposed entry -> original entry
        rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg)
        key = col.to(torch.int64) * n_rows + rows
        perm = torch.argsort(key, stable=True)
        self.perm = perm
        self.col_t = rows[perm].to(torch.int32).contiguous()
        cnt = torch.bincount(col.to(torch.int64), minlength=n_cols)
        crow_t = torch.zeros(n_cols + 1, dtype=torch.int64, device=dev)
        crow_t[1:] = torch.cumsum(cnt, 0)
        self.crow_t = crow_t.to(torch.int32).contiguous()
        self.short_rows_t, self.long_rows_t = self._split(cnt)
        self.pattern_symmetric = (n_rows == n_cols and torch.equal(self.crow_t, self.crow)
                                  and torch.equal(self.col_t, self.col))
        self._sym_vals_key = None
        self._sym_vals = False

    @staticmethod
    def _split(deg):
        hub = deg > HUB_DEGREE
        return (torch.nonzero(~hub).flatten().to(torch.int32).contiguous(),
                torch.nonzero(hub).flatten().to(torch.int32).contiguous())

    @staticmethod
    def _same_values(kept, kept_version, val) -> bool:
        """True when `val` is provably the tensor the cache was filled from.  The cache keeps a strong reference to
        that tensor, so its address cannot be handed to a new allocation while it is the key (a SparseDropout draw
        freed and re-allocated at the same address must miss); a tensor at the same address with the same size and
        version counter is then the kept tensor itself or a view / detach of it."""
        return (kept is not None and val.data_ptr() == kept.data_ptr() and val.numel() == kept.numel()
                and val.dtype == kept.dtype and val._version == kept_version)

    def transposed_values(self, val: torch.Tensor) -> torch.Tensor:
        if self.pattern_symmetric:
            if not self._same_values(self._sym_vals_key, self.__dict__.get("_sym_vals_version"), val):
                self._sym_vals = bool(torch.equal(val.index_select(0, self.perm), val))
                self._sym_vals_key, self._sym_vals_version = val, val._version
            if self._sym_vals:
                return val
        # the adjacency values are constants between SparseDropout draws: permute them once per tensor version
        if not self._same_values(self.__dict__.get("_tv_src"), self.__dict__.get("_tv_version"), val):
            self._tv = val.index_select(0, self.perm)
            self._tv_src, self._tv_version = val, val._version
        return self._tv


    # ---- tiled form (mi_spmm_tiled) ----
    TILES_PER_LAUNCH = 512          # about two workgroups per CU, each with about nnz / 512 edges
    COL_BLOCK_BYTES = 2 << 20       # rows of X per column block: half of an XCD's 4 MiB L2

    def tiles(self, D: int, transposed: bool):
        """The tile plan of A (or A^T) for rows of D floats, built once per (pattern, D): row tiles of about equal edge
        count and at most 64 KiB of LDS sums, every tile's edges grouped by column block.  None when the kernel cannot
        take the matrix (columns >= 2^23, D outside the float4 kernels)."""
        key = (int(D), bool(transposed))
        cache = self.__dict__.setdefault("_tiles", {})
        if key in cache:
            return cache[key]
        crow, col = (self.crow_t, self.col_t) if transposed else (self.crow, self.col)
        n_rows, n_cols = (self.shape[1], self.shape[0]) if transposed else self.shape
        plan = None
        if _float4_rows(D) and n_cols < (1 << 23) and self.nnz > 0 and n_rows > 0:
            dev = crow.device
            crow64 = crow.to(torch.int64)
            deg = crow64[1:] - crow64[:-1]
            budget = max(1, -(-self.nnz // self.TILES_PER_LAUNCH))
            max_rows = max(1, min(256, (64 * 1024) // (4 * D)))
            t_e = crow64[:-1] // budget                                  # tile id by edge budget (non-decreasing)
            rows = torch.arange(n_rows, device=dev)
            _, inv, cnt = torch.unique_consecutive(t_e, return_inverse=True, return_counts=True)
            grp_first = torch.cumsum(cnt, 0) - cnt                       # first row of each edge-budget group
            sub = (rows - grp_first[inv]) // max_rows                    # ... cut further by the row cap
            _, row_tile, tcnt = torch.unique_consecutive(inv * (n_rows // max_rows + 2) + sub, return_inverse=True,
                                                         return_counts=True)
            ntiles = int(tcnt.numel())
            tile_row0 = torch.zeros(ntiles + 1, dtype=torch.int64, device=dev)
            tile_row0[1:] = torch.cumsum(tcnt, 0)
            erow = torch.repeat_interleave(rows, deg)
            etile = row_tile[erow]
            lrow = erow - tile_row0[etile]
            cb_rows = max(1, self.COL_BLOCK_BYTES // (4 * D))
            col64 = col.to(torch.int64)
            nblocks = -(-n_cols // cb_rows)
            perm = torch.argsort(etile * nblocks + col64 // cb_rows, stable=True)
            ecr = (col64 | (lrow << 23))[perm].to(torch.int32).contiguous()
            tile_edge0 = torch.zeros(ntiles + 1, dtype=torch.int64, device=dev)
            tile_edge0[1:] = torch.cumsum(torch.bincount(etile, minlength=ntiles), 0)
            plan = dict(ntiles=ntiles, max_rows=int(tcnt.max()), ecr=ecr, perm=perm,
                        tile_edge0=tile_edge0.to(torch.int32).contiguous(), tile_row0=tile_row0.to(torch.int32).contiguous())
        cache[key] = plan
        return plan

    def tile_values(self, val: torch.Tensor, D: int, transposed: bool) -> torch.Tensor:
        """`val` (the values of A, or of A^T when transposed) in the tile plan's edge order; cached per values tensor
        (they only change with a SparseDropout draw)."""
        key = ("_tilev", int(D), bool(transposed))
        hit = self.__dict__.get(key)
        if hit is not None and self._same_values(hit[0], hit[1], val):
            return hit[2]
        out = val.index_select(0, self.tiles(D, transposed)["perm"])
        self.__dict__[key] = (val, val._version, out)
        return out


    # ---- task-balanced, slice-phased form (mi_spmm_sliced, round 4) ----
    SLICE_BYTES = int(_os_env("MI_SPMM_SLICE_KB", "2048")) << 10      # rows of X per column slice (default: half of an XCD's 4 MiB L2)
    TASK_NNZ = int(_os_env("MI_SPMM_TASK_NNZ", "512"))                # nonzeros per task: few enough tasks that all are resident at once
    GROUP_ROWS = 4                  # kGroupRows of csrc/spmm.hip: rows a lane group (narrow task) / a wide task owns
    WIDE_MIN = 48                   # rows with more nonzeros than this get a wide task (all lane groups stride them)

    def sliced(self, D: int, transposed: bool):
        """The task plan of A (or A^T) for rows of D floats, built once per (pattern, D) on the host (csrc/spmm.hip,
        k_spmm_sliced): NPW = 256 / D lane groups per wave.  Rows in row order; a row of <= WIDE_MIN nonzeros joins the open
        NARROW task (<= 4 NPW rows, <= TASK_NNZ nonzeros; row i of the task belongs to lane group i % NPW), a heavier one the
        open WIDE task (<= 4 rows, <= TASK_NNZ nonzeros unless alone), hubs (> HUB_DEGREE) are listed apart.  Inside an
        owner's range the edges are sorted by column slice.  None when the kernel cannot take the matrix."""
        key = (int(D), bool(transposed))
        cache = self.__dict__.setdefault("_sliced", {})
        if key in cache:
            return cache[key]
        import numpy as np

        crow_d, col_d = (self.crow_t, self.col_t) if transposed else (self.crow, self.col)
        n_rows, n_cols = (self.shape[1], self.shape[0]) if transposed else self.shape
        plan = None
        if _float4_rows(D) and n_cols < (1 << 28) and self.nnz > 0 and n_rows > 0:
            dev = crow_d.device
            NPW, GR = 256 // D, self.GROUP_ROWS
            crow = crow_d.cpu().numpy().astype(np.int64)
            col = col_d.cpu().numpy().astype(np.int64)
            deg = crow[1:] - crow[:-1]
            hub = deg > HUB_DEGREE
            slice_rows = max(1, self.SLICE_BYTES // (4 * D))
            task_of = np.full(n_rows, -1, dtype=np.int64)
            slot_of = np.zeros(n_rows, dtype=np.int64)       # lane group (narrow) / 0 (wide)
            j_of = np.zeros(n_rows, dtype=np.int64)          # first or second row of its owner
            wide_flags = []
            open_n = None         # [task id, rows so far, nonzeros so far] of the open narrow / wide task
            open_w = None
            T = 0
            degl, hubl = deg.tolist(), hub.tolist()
            for r in range(n_rows):
                if hubl[r]:
                    continue
                d = degl[r]
                if d > self.WIDE_MIN:
                    if open_w is None or open_w[1] == GR or open_w[2] + d > self.TASK_NNZ:
                        open_w = [T, 0, 0]
                        wide_flags.append(1)
                        T += 1
                    task_of[r], slot_of[r], j_of[r] = open_w[0], 0, open_w[1]
                    open_w[1] += 1
                    open_w[2] += d
                else:
                    if open_n is None or open_n[1] == GR * NPW or open_n[2] + d > self.TASK_NNZ:
                        open_n = [T, 0, 0]
                        wide_flags.append(0)
                        T += 1
                    task_of[r], slot_of[r], j_of[r] = open_n[0], open_n[1] % NPW, open_n[1] // NPW
                    open_n[1] += 1
                    open_n[2] += d
            twide = np.asarray(wide_flags, dtype=np.uint8)
            erow = np.repeat(np.arange(n_rows), deg)
            keep = np.nonzero(~hub[erow])[0]
            er = erow[keep]
            nsl = -(-n_cols // slice_rows)
            # owner = (task, lane group); inside an owner: slice, then first / second row, then the CSR's column order
            ek = ((task_of[er] * NPW + slot_of[er]) * nsl + col[keep] // slice_rows) * GR + j_of[er]
            order = np.argsort(ek, kind="stable")
            perm = keep[order]
            ecol = col[perm] | (j_of[erow[perm]] << 28)
            ecol = np.where(ecol >= 2 ** 31, ecol - 2 ** 32, ecol).astype(np.int32)       # bit pattern of the uint32 word
            owner = task_of[er] * NPW + slot_of[er]
            cnt = np.bincount(owner, minlength=max(T, 1) * NPW)
            starts = np.zeros(max(T, 1) * NPW + 1, dtype=np.int64)
            starts[1:] = np.cumsum(cnt)
            tptr = np.zeros((max(T, 1), NPW + 1), dtype=np.int64)
            tptr[:, :NPW] = starts[:-1].reshape(-1, NPW)
            tptr[:, NPW] = starts[NPW::NPW]
            # a wide task keeps all its edges under lane group 0: its range is [tptr[0], tptr[1]) = group 0's, as the kernel reads it
            trows = np.full((max(T, 1), GR * NPW), -1, dtype=np.int32)
            nh = np.nonzero(~hub)[0]
            trows[task_of[nh], j_of[nh] * NPW + slot_of[nh]] = nh
            plan = dict(T=T, NPW=NPW, slice_rows=int(slice_rows), tptr=torch.from_numpy(tptr.astype(np.int32)).to(dev),
                        trows=torch.from_numpy(trows).to(dev), twide=torch.from_numpy(twide if T else np.zeros(1, np.uint8)).to(dev),
                        ecol=torch.from_numpy(ecol).to(dev), perm=torch.from_numpy(perm.astype(np.int64)).to(dev),
                        long_rows=torch.from_numpy(np.nonzero(hub)[0].astype(np.int32)).to(dev))
        cache[key] = plan
        return plan

    def sliced_values(self, val: torch.Tensor, D: int, transposed: bool) -> torch.Tensor:
        """`val` (the values of A, or of A^T when transposed) in the task plan's edge order; cached per values tensor."""
        key = ("_slicedv", int(D), bool(transposed))
        hit = self.__dict__.get(key)
        if hit is not None and self._same_values(hit[0], hit[1], val):
            return hit[2]
        out = val.index_select(0, self.sliced(D, transposed)["perm"])
        self.__dict__[key] = (val, val._version, out)
        return out


# The tiled SpMM (csrc/spmm.hip, round 3: row tiles whose sums live in LDS, edges walked column block by column block,
# one edge per 16-lane group, LDS float adds) is parity-green and OPT-IN (MI_SPMM_TILED=1): on MI355X it runs 13x SLOWER
# than the row-per-wave kernels (1073 vs 80 us per Yelp2018-shaped layer) because ds_add_f32 executes a wave's 64 lanes
# one after the other — 170 cycles per wave-instruction whatever the address pattern, against 5-8 for ds_add_u32 and
# 10-15 for a plain read + add + write (tools/probe_lds_atomic.hip, profiles/r03_lds_atomic_probe.txt).
import os as _os

TILED_SPMM = _os.environ.get("MI_SPMM_TILED", "0") == "1"
# The slice-phased SpMM (round 4, mi_spmm_sliced) is parity-green and OPT-IN: "0" (default) = never, "1" = for operands
# larger than an XCD's L2, "2" = always (tests).  Measured on MI355X at the Yelp2018 shape (profiles/r04_spmm_counters.txt):
# L2 hit rate 0.38 -> 0.50 and 12 % fewer fabric-side bytes, but 70-82 us per layer against 75 for the row-per-wave kernel
# whatever the slice size (128 KiB ... 4 MiB) and whether the tasks run in one round or three: the waves do not stay in phase,
# and what bounds both kernels is the rate at which the cache hierarchy serves random 256-B rows (~8 TB/s), not its hit rate.
SLICED_SPMM = int(_os.environ.get("MI_SPMM_SLICED", "0"))
SLICED_MIN_BYTES = 4 << 20

_plans = {}


def csr_plan(matrix: torch.Tensor) -> CsrPlan:
    crow, col = matrix.crow_indices(), matrix.col_indices()
    key = (crow.data_ptr(), col.data_ptr(), col.numel(), tuple(matrix.shape), str(crow.device))
    plan = _plans.get(key)
    if plan is not None and (plan.src_versions != (crow._version, col._version)):
        plan = None     # the index tensors were written in place since the plan was built
    if plan is None:
        if len(_plans) > 16:
            _plans.clear()
        plan = CsrPlan(crow, col, tuple(matrix.shape))
        _plans[key] = plan
    return plan


def _spmm(plan: CsrPlan, transposed: bool, val, Xa, Xb, x_split, Y, acc_a, acc_b, acc_split, acc_out, scale, D, xmask=None,
          rows=None):
    """xmask (optional, int32 words, one bit per row of X): rows whose bit is 0 are all zeros and are not fetched.
    rows (optional, (short int32 list, hub int32 list, hub_need bytes)): only these OUTPUT rows are computed — of the hubs
    those whose hub_need byte is set (repeats allowed: the outputs must then not alias the inputs); the other rows of Y /
    acc_out are left untouched."""
    lib = _lib.load()
    n_x = plan.shape[0] if transposed else plan.shape[1]
    if rows is None and (SLICED_SPMM == 2 or (SLICED_SPMM == 1 and n_x * D * 4 > SLICED_MIN_BYTES)) and not TILED_SPMM:
        sp = plan.sliced(D, transposed)
        if sp is not None:
            crow_h, col_h = (plan.crow_t, plan.col_t) if transposed else (plan.crow, plan.col)
            ev = plan.sliced_values(val, D, transposed)
            lr = sp["long_rows"]
            _lib.check(
                lib.mi_spmm_sliced(crow_h.data_ptr(), col_h.data_ptr(), val.data_ptr(), sp["tptr"].data_ptr(), sp["trows"].data_ptr(),
                                   sp["twide"].data_ptr(), sp["ecol"].data_ptr(), ev.data_ptr(), sp["T"], Xa.data_ptr(), _lib.ptr(Xb), x_split,
                                   _lib.ptr(Y), _lib.ptr(acc_a), _lib.ptr(acc_b), acc_split, _lib.ptr(acc_out), float(scale), D,
                                   lr.data_ptr() if lr.numel() else None, lr.numel(), _lib.ptr(xmask), _lib.stream_ptr(val.device)),
                "mi_spmm_sliced")
            return
    tp = plan.tiles(D, transposed) if (TILED_SPMM and not DETERMINISTIC and rows is None) else None
    if tp is not None:
        ev = plan.tile_values(val, D, transposed)
        _lib.check(
            lib.mi_spmm_tiled(tp["tile_edge0"].data_ptr(), tp["tile_row0"].data_ptr(), tp["ntiles"], tp["max_rows"],
                              tp["ecr"].data_ptr(), ev.data_ptr(), Xa.data_ptr(), _lib.ptr(Xb), x_split, _lib.ptr(Y),
                              _lib.ptr(acc_a), _lib.ptr(acc_b), acc_split, _lib.ptr(acc_out), float(scale), D,
                              _lib.stream_ptr(val.device)),
            "mi_spmm_tiled",
        )
        return
    if transposed:
        crow, col, sr, lr, n_rows = plan.crow_t, plan.col_t, plan.short_rows_t, plan.long_rows_t, plan.shape[1]
    else:
        crow, col, sr, lr, n_rows = plan.crow, plan.col, plan.short_rows, plan.long_rows, plan.shape[0]
    need = None
    if rows is not None:
        sr, lr, need = rows
    dev = val.device
    _lib.check(
        lib.mi_spmm_csr_sel(crow.data_ptr(), col.data_ptr(), val.data_ptr(), Xa.data_ptr(), _lib.ptr(Xb), x_split,
                            _lib.ptr(Y), _lib.ptr(acc_a), _lib.ptr(acc_b), acc_split, _lib.ptr(acc_out), float(scale),
                            n_rows, D, sr.data_ptr() if sr.numel() else None, sr.numel(),
                            lr.data_ptr() if lr.numel() else None, lr.numel(), _lib.ptr(xmask), _lib.ptr(need),
                            _lib.stream_ptr(dev)),
        "mi_spmm_csr_sel",
    )


MASK_FIRST_BACKWARD_LAYER = _os.environ.get("MI_SPMM_ROW_MASK", "1") == "1"


def _row_mask(Xa, Xb, D):
    """Bit per row of [Xa; Xb]: 1 = the row has a non-zero element (mi_row_mask), or None where that kernel has no form."""
    lpr = D // 4
    if D % 4 or lpr > 64 or lpr & (lpr - 1):
        return None
    n = Xa.shape[0] + (Xb.shape[0] if Xb is not None else 0)
    mask = torch.empty(((n + 31) // 32,), dtype=torch.int32, device=Xa.device)
    _lib.check(_lib.load().mi_row_mask(Xa.data_ptr(), _lib.ptr(Xb), Xa.shape[0] if Xb is not None else 0, n, D, mask.data_ptr(),
                                       _lib.stream_ptr(Xa.device)), "mi_row_mask")
    return mask


def _propagate(plan, transposed, val, Xa, Xb, num_layers, sparse_input=False, last_rows=None):
    """res = (sum_{k=0..L} A^k X) / (L+1) with X = [Xa; Xb] (Xb may be None).  sparse_input: X is expected to be zero on
    most rows (the gradient entering a backward propagation is non-zero only on the batch's rows): a row mask of X is
    built on the device (one pass over X) and the first layer does not fetch the zero rows — same result.
    last_rows ((users, pos, neg, item_base): int64 ids, row = user id / item_base + item id; forward only): the caller reads
    the result at these rows only, so the LAST layer computes only them; every other row of the result is undefined."""
    dev = Xa.device
    D = Xa.shape[1]
    n = Xa.shape[0] + (Xb.shape[0] if Xb is not None else 0)
    x_split = Xa.shape[0] if Xb is not None else 0
    if num_layers == 0:
        return torch.cat([Xa, Xb]) if Xb is not None else Xa.clone()
    acc = torch.empty((n, D), dtype=torch.float32, device=dev)
    bufs = [torch.empty((n, D), dtype=torch.float32, device=dev) for _ in range(min(2, num_layers - 1))]
    cur_a, cur_b, cur_split = Xa, Xb, x_split
    for k in range(1, num_layers + 1):
        last = k == num_layers
        Y = None if last else bufs[(k - 1) % len(bufs)]
        scale = 1.0 / (num_layers + 1) if last else 1.0
        rows, out = None, acc
        if last and last_rows is not None and not transposed and plan.filler_row >= 0 and _float4_rows(D):
            # one launch turns the batch into the one-wave-per-row list (hubs -> a filler row + their hub_need flag)
            users, pos, neg, item_base = last_rows
            short = torch.empty((3 * users.numel(),), dtype=torch.int32, device=dev)
            _lib.check(_lib.load().mi_batch_row_list(users.data_ptr(), pos.data_ptr(), neg.data_ptr(), users.numel(),
                                                     item_base, n, plan.is_hub.data_ptr(),
                                                     plan.filler_row, short.data_ptr(), plan.hub_need.data_ptr(),
                                                     _lib.stream_ptr(dev)), "mi_batch_row_list")
            rows = (short, plan.long_rows, plan.hub_need)
            out = torch.empty_like(acc)          # repeats in the list: the result must not be updated in place
        if k == 1:
            xmask = _row_mask(Xa, Xb, D) if (sparse_input and MASK_FIRST_BACKWARD_LAYER) else None
            _spmm(plan, transposed, val, cur_a, cur_b, cur_split, Y, Xa, Xb, x_split, out, scale, D, xmask=xmask, rows=rows)
        else:
            _spmm(plan, transposed, val, cur_a, None, 0, Y, acc, None, 0, out, scale, D, rows=rows)
        acc = out
        cur_a, cur_b, cur_split = Y, None, 0
    return acc


class LightGCNPropagate(torch.autograd.Function):
    """(sum_{k=0..L} A^k [Xa;Xb]) / (L+1) — src/models/lightgcn.py:79-87.  Backward is the same
    propagation with A^T applied to the incoming gradient."""

    @staticmethod
    def forward(ctx, val, Xa, Xb, plan, num_layers: int):
        _lib.require_gpu(val, Xa, Xb)
        if val.requires_grad:
            raise NotImplementedError("gradients w.r.t. the adjacency values are not provided")
        if plan.shape[0] != plan.shape[1]:
            raise ValueError("propagation needs a square adjacency")
        valc, Xac = _f32c(val), _f32c(Xa)
        Xbc = None if Xb is None else _f32c(Xb)
        n = Xac.shape[0] + (Xbc.shape[0] if Xbc is not None else 0)
        if n != plan.shape[0]:
            raise ValueError(f"adjacency is {plan.shape} but the embedding tables have {n} rows")
        ctx.plan, ctx.num_layers = plan, num_layers
        ctx.split = Xac.shape[0] if Xbc is not None else None
        ctx.save_for_backward(valc)
        res = _propagate(plan, False, valc, Xac, Xbc, num_layers)
        if ctx.split is None:
            return res
        # two tables in, two tables out (LightGCN splits the result anyway, src/models/lightgcn.py:87): the backward
        # then receives the two gradients separately and reads them as two row segments — no cat, no split
        return res[: ctx.split], res[ctx.split:]

    @staticmethod
    def backward(ctx, *gs):
        (valc,) = ctx.saved_tensors
        plan = ctx.plan
        val_t = plan.transposed_values(valc)
        if ctx.split is None:
            g = _f32c(gs[0])
            gX = _propagate(plan, True, val_t, g, None, ctx.num_layers, sparse_input=True)
            return None, gX, None, None, None
        ga, gb = gs
        n_a, n_b = ctx.split, plan.shape[0] - ctx.split
        D = (ga if ga is not None else gb).shape[1]
        dev = valc.device
        ga = _f32c(ga) if ga is not None else torch.zeros((n_a, D), dtype=torch.float32, device=dev)
        gb = _f32c(gb) if gb is not None else torch.zeros((n_b, D), dtype=torch.float32, device=dev)
        gX = _propagate(plan, True, val_t, ga, gb, ctx.num_layers, sparse_input=True)
        return None, gX[: ctx.split], gX[ctx.split:], None, None


class LightGCNPropagateReg(torch.autograd.Function):
    """LightGCNPropagate of plain tables AND get_reg_loss over a batch's rows of those tables
    (src/models/lightgcn.py:79-100,166-173) as one node: (res_a, res_b, reg) for two tables, (res, reg) for one table
    (Xb None: SingleLightGCN, items follow the `item_base` users).  The point is the backward: the regulariser's
    gradient touches 3 B rows, and as a node of its own it reaches the tables as zero-filled [N, D] tensors that
    autograd then adds to the propagation's gradients (two fills + two adds of 8 - 10 MB each per step at Yelp2018 size);
    here mi_rowsq_bwd adds those rows straight into the propagation's gradient."""

    @staticmethod
    def forward(ctx, val, Xa, Xb, plan, num_layers: int, users, pos, neg, batch_rows_only: bool = False, item_base: int = 0):
        dev = _lib.require_gpu(val, Xa, Xb, users)
        if val.requires_grad:
            raise NotImplementedError("gradients w.r.t. the adjacency values are not provided")
        valc, Xac = _f32c(val), _f32c(Xa)
        Xbc = None if Xb is None else _f32c(Xb)
        n = Xac.shape[0] + (Xbc.shape[0] if Xbc is not None else 0)
        if n != plan.shape[0] or plan.shape[0] != plan.shape[1]:
            raise ValueError(f"adjacency is {plan.shape} but the embedding tables have {n} rows")
        ui, pi, ni = (_i64c(t).view(-1) for t in (users, pos, neg))
        B, D = ui.numel(), Xac.shape[1]
        if pi.numel() != B or ni.numel() != B or B == 0:
            raise ValueError("reg loss: users / positives / negatives must be [B] indices")
        lib = _lib.load()
        if Xbc is None:              # one table: the item rows start at item_base
            item_tab, pr, nr = Xac, pi + item_base, ni + item_base
        else:
            item_tab, pr, nr, item_base = Xbc, pi, ni, Xac.shape[0]
        last_rows = (ui, pi, ni, item_base) if batch_rows_only else None
        res = _propagate(plan, False, valc, Xac, Xbc, num_layers, last_rows=last_rows)
        from .losses import _ticket_workspace

        ws, armed = _ticket_workspace("rowsq", dev, lib.mi_bpr_workspace_elems(B))
        reg = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check((lib.mi_rowsq_fwd_armed if armed else lib.mi_rowsq_fwd)(Xac.data_ptr(), ui.data_ptr(), item_tab.data_ptr(), pr.data_ptr(), item_tab.data_ptr(),
                                    nr.data_ptr(), B, D, Xac.shape[0], item_tab.shape[0], item_tab.shape[0],
                                    _lib.err_word(dev).data_ptr(), ws.data_ptr(), reg.data_ptr(), _lib.stream_ptr(dev)),
                   "mi_rowsq_fwd")
        ctx.plan, ctx.num_layers = plan, num_layers
        ctx.split = None if Xbc is None else Xac.shape[0]
        ctx.save_for_backward(valc, Xac, item_tab, ui, pr, nr)
        if ctx.split is None:
            return res, reg.view(())
        return res[: ctx.split], res[ctx.split:], reg.view(())

    @staticmethod
    def backward(ctx, *gs):
        valc, Xac, item_tab, ui, pr, nr = ctx.saved_tensors
        plan, split = ctx.plan, ctx.split
        dev, D = valc.device, Xac.shape[1]
        greg = gs[-1]
        val_t = plan.transposed_values(valc)
        if split is None:
            g = gs[0]
            gX = (_propagate(plan, True, val_t, _f32c(g), None, ctx.num_layers, sparse_input=True) if g is not None
                  else torch.zeros((plan.shape[0], D), dtype=torch.float32, device=dev))
            gXa = gXb = gX
        else:
            ga, gb = gs[0], gs[1]
            if ga is None and gb is None:
                gX = torch.zeros((plan.shape[0], D), dtype=torch.float32, device=dev)
            else:
                ga = _f32c(ga) if ga is not None else torch.zeros((split, D), dtype=torch.float32, device=dev)
                gb = _f32c(gb) if gb is not None else torch.zeros((plan.shape[0] - split, D), dtype=torch.float32, device=dev)
                gX = _propagate(plan, True, val_t, ga, gb, ctx.num_layers, sparse_input=True)
            gXa, gXb = gX[:split], gX[split:]
        if greg is not None:
            g = _f32c(greg).view(1)
            _lib.check(_lib.load().mi_rowsq_bwd(Xac.data_ptr(), ui.data_ptr(), item_tab.data_ptr(), pr.data_ptr(),
                                                item_tab.data_ptr(), nr.data_ptr(), ui.numel(), D, Xac.shape[0],
                                                item_tab.shape[0], item_tab.shape[0], g.data_ptr(), gXa.data_ptr(),
                                                gXb.data_ptr(), gXb.data_ptr(), _lib.stream_ptr(dev)), "mi_rowsq_bwd")
        if split is None:
            return None, gX, None, None, None, None, None, None, None, None
        return None, gXa, gXb, None, None, None, None, None, None, None


def lightgcn_propagate_reg(matrix: torch.Tensor, Xa: torch.Tensor, Xb: Optional[torch.Tensor], num_layers: int, users, pos, neg,
                           batch_rows_only: bool = False, item_base: int = 0):
    """(all_user_emb, all_item_emb, reg_loss), or (all_emb, reg_loss) for ONE table (Xb None; item row = item_base + item id)
    — see LightGCNPropagateReg.  batch_rows_only: the result is valid ONLY at rows `users` / `pos`, `neg` (what BPR and
    InfoNCE over the batch read): the last propagation layer computes just those rows instead of all of them."""
    if matrix.layout != torch.sparse_csr:
        if matrix.layout == torch.sparse_coo:
            matrix = matrix.coalesce().to_sparse_csr()
        else:
            raise ValueError(f"Not supported matrix layout: {matrix.layout}")
    return LightGCNPropagateReg.apply(matrix.values(), Xa, Xb, csr_plan(matrix), num_layers, users, pos, neg, batch_rows_only,
                                      int(item_base))


def lightgcn_propagate(matrix: torch.Tensor, Xa: torch.Tensor, Xb: Optional[torch.Tensor], num_layers: int):
    if matrix.layout != torch.sparse_csr:
        if matrix.layout == torch.sparse_coo:
            matrix = matrix.coalesce().to_sparse_csr()
        else:
            raise ValueError(f"Not supported matrix layout: {matrix.layout}")
    plan = csr_plan(matrix)
    return LightGCNPropagate.apply(matrix.values(), Xa, Xb, plan, num_layers)


def spmm(matrix: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """matrix @ X for a sparse-CSR matrix (rectangular allowed; src/models/hccf.py:56-57 shape)."""
    return SpMM.apply(matrix.values(), X, csr_plan(matrix))


class SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, val, X, plan):
        _lib.require_gpu(val, X)
        valc, Xc = _f32c(val), _f32c(X)
        if Xc.shape[0] != plan.shape[1]:
            raise ValueError("shape mismatch in sparse @ dense")
        Y = torch.empty((plan.shape[0], Xc.shape[1]), dtype=torch.float32, device=Xc.device)
        _spmm(plan, False, valc, Xc, None, 0, Y, None, None, 0, None, 1.0, Xc.shape[1])
        ctx.plan = plan
        ctx.save_for_backward(valc)
        return Y

    @staticmethod
    def backward(ctx, g):
        (valc,) = ctx.saved_tensors
        plan = ctx.plan
        g = _f32c(g)
        gX = torch.empty((plan.shape[1], g.shape[1]), dtype=torch.float32, device=g.device)
        _spmm(plan, True, plan.transposed_values(valc), g, None, 0, gX, None, None, 0, None, 1.0, g.shape[1])
        return None, gX, None


# --------------------------------------------------------------------------------------
# fp32 MFMA GEMM with fused epilogues (building block of the CrossNet heads)
EPI = {"none": 0, "bias": 1, "tanh": 2, "cross": 3, "add": 4, "mul_dtanh": 5, "tanh_gate": 6, "accum": 7}


def gemm(A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, batch=1, sA=0, sB=0, sC=0,
         kgroups=1, gA=0, gB=0, epi="none", bias=None, R1=None, ldr1=0, sR1=0, R2=None, ldr2=0, sR2=0,
         rowscale=None, nrs=0, C2=None, ldc2=0, sC2=0, splitk=0):
    """Raw launcher of mi_gemm_f32 over torch buffers (pointers may be offset views).

    splitk=0 picks automatically: when the output is only a few 64x64 tiles but K is long (the
    weight-gradient shapes) the K loop is spread over enough workgroups to fill the 256 CUs."""
    dev = _lib.require_gpu(A, B, C)
    if splitk == 0:
        splitk = 1
        if epi in ("none", "accum"):
            tiles = -(-M // 64) * -(-N // 64) * batch
            ksteps = -(-K // 32) * kgroups
            if tiles < 256 and ksteps >= 16 and not DETERMINISTIC:   # slices meet in float atomics
                splitk = max(1, min(ksteps // 4, -(-512 // tiles), 65535 // max(batch, 1)))
        if splitk > 1 and epi == "none":
            if C.is_contiguous() and C.numel() == batch * M * N:
                C.zero_()
            else:
                _zero_strided(C, M, N, ldc, batch, sC)
    _lib.check(
        _lib.load().mi_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, lda, ldb, ldc, int(transA),
                                int(transB), batch, sA, sB, sC, kgroups, gA, gB, EPI[epi], _lib.ptr(bias),
                                _lib.ptr(R1), ldr1, sR1, _lib.ptr(R2), ldr2, sR2, _lib.ptr(rowscale), nrs,
                                _lib.ptr(C2), ldc2, sC2, splitk, _lib.stream_ptr(dev)),
        "mi_gemm_f32",
    )
    return C


class _GemmProblem(_ctypes.Structure):      # mi_gemm_problem (include/mi355x_recsys.h)
    _fields_ = [("A", _ctypes.c_void_p), ("B", _ctypes.c_void_p), ("C", _ctypes.c_void_p),
                ("M", _ctypes.c_int32), ("N", _ctypes.c_int32), ("K", _ctypes.c_int32),
                ("lda", _ctypes.c_int32), ("ldb", _ctypes.c_int32), ("ldc", _ctypes.c_int32), ("batch", _ctypes.c_int32),
                ("sA", _ctypes.c_int64), ("sB", _ctypes.c_int64), ("sC", _ctypes.c_int64),
                ("splitk", _ctypes.c_int32), ("accumulate", _ctypes.c_int32)]


MAX_GEMM_PROBLEMS = 16


def auto_splitk(M, N, K, batch=1):
    """K-slices that fill the 256 CUs when the output is only a few 64x64 tiles (1 in deterministic mode: the slices meet
    in float atomics)."""
    tiles = -(-M // 64) * -(-N // 64) * batch
    ksteps = -(-K // 32)
    if tiles < 256 and ksteps >= 16 and not DETERMINISTIC:
        return max(1, min(ksteps // 4, -(-512 // tiles), 65535 // max(batch, 1)))
    return 1


def _multi_problem_array(chunk):
    """ctypes array of mi_gemm_problem for one launch.  splitk: the caller's value, else 0 = the library cuts the launch
    (gemm.hip auto_splitk: the slice count that loads the CUs most evenly, atomics priced in);
    deterministic mode never splits (slices meet in float atomics)."""
    arr = (_GemmProblem * len(chunk))()
    for d, q in zip(arr, chunk):
        batch = q.get("batch", 1)
        d.A, d.B, d.C = q["A"].data_ptr(), q["B"].data_ptr(), q["C"].data_ptr()
        d.M, d.N, d.K = q["M"], q["N"], q["K"]
        d.lda, d.ldb, d.ldc, d.batch = q["lda"], q["ldb"], q["ldc"], batch
        d.sA, d.sB, d.sC = q.get("sA", 0), q.get("sB", 0), q.get("sC", 0)
        d.splitk = q.get("splitk", 0) or (1 if DETERMINISTIC else 0)
        d.accumulate = int(bool(q.get("accumulate", False)))
    return arr


def gemm_multi_plan(problems, transA=False, transB=False):
    """(kind, workgroups, splitk[n]) of the launch gemm_multi would make for <= 16 problems (mi_gemm_f32_multi_plan: host
    arithmetic, no GPU work); kind 1 = the LDS-DMA kernel of the weight-gradient form, 0 = the general kernel."""
    arr = _multi_problem_array(problems)
    kind, wgs = _ctypes.c_int32(), _ctypes.c_int64()
    splitk = (_ctypes.c_int32 * max(1, len(problems)))()
    _lib.check(_lib.load().mi_gemm_f32_multi_plan(_ctypes.addressof(arr), len(problems), int(transA), int(transB),
                                                  _ctypes.addressof(kind), _ctypes.addressof(wgs), _ctypes.addressof(splitk)),
               "mi_gemm_f32_multi_plan")
    return kind.value, wgs.value, list(splitk[:len(problems)])


class PrefetchRowsJob(_ctypes.Structure):
    """mi_prefetch_rows_job: the NEXT batch's ids and the tables its lookup will read."""
    _fields_ = [("idx", _ctypes.c_void_p), ("offsets", _ctypes.c_void_p), ("W", _ctypes.c_void_p), ("w1", _ctypes.c_void_p),
                ("ldw", _ctypes.c_int64), ("ldw1", _ctypes.c_int64), ("B", _ctypes.c_int64), ("N", _ctypes.c_int64),
                ("F", _ctypes.c_int32)]


def gemm_multi(problems, transA=False, transB=False, ride: Optional[PrefetchRowsJob] = None):
    """problems: dicts with A, B, C (torch buffers / views), M, N, K, lda, ldb, ldc and optionally batch, sA, sB, sC, splitk
    (0 / absent = the library's cut), accumulate — all the same operand layout, independent of each other: ONE launch per 16
    problems (mi_gemm_f32_multi).  C of a problem must be zero (or accumulate) unless splitk == 1: the caller allocates
    them zero-filled.  ride: a prefetch job the (first) launch carries in extra workgroups (mi_gemm_f32_multi_ride)."""
    if not problems:
        if ride is not None:
            _lib.check(_lib.load().mi_gemm_f32_multi_ride(None, 0, int(transA), int(transB), _ctypes.byref(ride),
                                                          _lib.stream_ptr(torch.device("cuda", torch.cuda.current_device()))),
                       "mi_gemm_f32_multi_ride")
        return
    dev = _lib.require_gpu(*[q[k] for q in problems for k in ("A", "B", "C")])
    lib, stream = _lib.load(), _lib.stream_ptr(dev)
    for i in range(0, len(problems), MAX_GEMM_PROBLEMS):
        chunk = problems[i:i + MAX_GEMM_PROBLEMS]
        arr = _multi_problem_array(chunk)
        _lib.check(lib.mi_gemm_f32_multi_ride(_ctypes.addressof(arr), len(chunk), int(transA), int(transB),
                                              _ctypes.byref(ride) if (ride is not None and i == 0) else None, stream),
                   "mi_gemm_f32_multi_ride")


PANEL_GEMM = True        # False: every product stays on the 64x64-tile kernel (mi_gemm_f32)


def gemm_panel(A, lda, B, ldb, b_layout, C, ldc, M, N, K, gw=None, gstride=0, epi="none", bias=None, R1=None, R2=None,
               rowscale=None, nrs=0, C2=None) -> bool:
    """mi_gemm_f32_panel over torch buffers; returns False — nothing launched — when the shape / alignment is outside that
    kernel, so the caller issues its mi_gemm_f32 form instead."""
    if gw is None:
        gw = K if b_layout == 0 else N
    if not PANEL_GEMM or M == 0 or any(v % 4 for v in (lda, ldb, ldc, N, K, gw, gstride)) or epi not in ("none", "tanh", "cross", "add"):
        return False
    if any(t is not None and t.data_ptr() % 16 for t in (A, B, C, bias, R1, R2, C2)):
        return False
    dev = _lib.require_gpu(A, B, C)
    _lib.check(_lib.load().mi_gemm_f32_panel(A.data_ptr(), lda, B.data_ptr(), ldb, b_layout, gw, gstride, C.data_ptr(), ldc, M, N,
                                             K, EPI[epi], _lib.ptr(bias), _lib.ptr(R1), _lib.ptr(R2), _lib.ptr(rowscale), nrs,
                                             _lib.ptr(C2), _lib.stream_ptr(dev)), "mi_gemm_f32_panel")
    return True


def _zero_strided(C, M, N, ldc, batch, sC):
    for z in range(batch):
        torch.as_strided(C, (M, N), (ldc, 1), C.storage_offset() + z * sC).zero_()


# --------------------------------------------------------------------------------------
# TT-Rec lookup
import ctypes as _ct  # noqa: E402


def _tt_host_args(cores, p_shapes, q_shapes, ranks):
    n = len(cores)
    ptrs = (_ct.c_void_p * n)(*[c.data_ptr() for c in cores])
    return (ptrs, (_ct.c_int32 * n)(*p_shapes), (_ct.c_int32 * n)(*q_shapes), (_ct.c_int32 * (n + 1))(*ranks))


class TTLookup(torch.autograd.Function):
    """out[n, D] = TT-Rec rows (tensortrain_embeddings.py:128-150); dense core grads via atomics."""

    @staticmethod
    def forward(ctx, idx, num_item: int, p_shapes, q_shapes, ranks, *cores):
        dev = _lib.require_gpu(idx, *cores)
        idxc = _i64c(idx).view(-1)
        cs = [_f32c(c) for c in cores]
        D = 1
        for q in q_shapes:
            D *= q
        out = torch.empty((idxc.numel(), D), dtype=torch.float32, device=dev)
        ptrs, p, q, r = _tt_host_args(cs, p_shapes, q_shapes, ranks)
        _lib.check(
            _lib.load().mi_tt_fwd(idxc.data_ptr(), ptrs, len(cs), p, q, r, out.data_ptr(), idxc.numel(), D,
                                  num_item, _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
            "mi_tt_fwd",
        )
        ctx.save_for_backward(idxc, *cs)
        ctx.meta = (num_item, list(p_shapes), list(q_shapes), list(ranks), D)
        return out

    @staticmethod
    def backward(ctx, g):
        idxc, *cs = ctx.saved_tensors
        num_item, p_shapes, q_shapes, ranks, D = ctx.meta
        g = _f32c(g)
        gcs = [torch.zeros_like(c) for c in cs]
        ptrs, p, q, r = _tt_host_args(cs, p_shapes, q_shapes, ranks)
        gptrs = (_ct.c_void_p * len(gcs))(*[c.data_ptr() for c in gcs])
        _lib.check(
            _lib.load().mi_tt_bwd(idxc.data_ptr(), g.data_ptr(), ptrs, gptrs, len(cs), p, q, r, idxc.numel(), D,
                                  num_item, _lib.stream_ptr(g.device)),
            "mi_tt_bwd",
        )
        return (None, None, None, None, None, *gcs)


# ---- grouped form: one MFMA GEMM per core slice (csrc/tt_grouped.hip, gemm.hip) ------------------
_TT_TILE = 64            # rows per GEMM tile: the rows of a digit group are padded to whole tiles
_TT_SEG = 2048           # reduction rows per slice-gradient segment (long groups are cut, float atomics join)
_TT_SEG0 = 256           # rows per segment of core 0's gradient sum
_TT_GROUPED_MIN = 4096   # below this many lookups the sort / planning launches cost more than they save
_TT_MAX_TILES = 65535    # gridDim.y
_TT_PLAN_MAX_P = 4096    # digits per level the device planner handles (beyond: torch sort / searchsorted)


class _TTLevelPlan:
    """Layout of one level: the lookups ordered by `digit`, each owning `H` consecutive rows, every digit
    group padded to whole 64-row tiles.  Built with device-side torch ops only (sort, searchsorted,
    cumsum): no host sync, static shapes."""

    def __init__(self, digit: torch.Tensor, H: int, p: int, seg: int):
        dev, n = digit.device, digit.numel()
        if p <= _TT_PLAN_MAX_P:
            # counting sort on the device: 3 launches instead of ~25 torch ops
            self.ntiles = (n * H + _TT_TILE - 1) // _TT_TILE + p
            self.mpad = self.ntiles * _TT_TILE
            self.nseg = (n * H + seg - 1) // seg + p
            ws = torch.empty(2 * p, dtype=torch.int32, device=dev)
            pbeg = torch.empty(p, dtype=torch.int64, device=dev)
            self.pos = torch.empty(n, dtype=torch.int64, device=dev)
            self.mtile_b = torch.empty(self.ntiles, dtype=torch.int32, device=dev)
            self.kseg = torch.empty((self.nseg, 3), dtype=torch.int64, device=dev)
            _lib.check(_lib.load().mi_tt_plan_level(digit.data_ptr(), n, p, H, seg, ws.data_ptr(), pbeg.data_ptr(),
                                                    self.pos.data_ptr(), self.mtile_b.data_ptr(), self.ntiles,
                                                    self.kseg.data_ptr(), self.nseg, _lib.stream_ptr(dev)),
                       "mi_tt_plan_level")
            return
        sorted_d, perm = torch.sort(digit, stable=True)
        bounds = torch.searchsorted(sorted_d, torch.arange(p + 1, device=dev, dtype=digit.dtype))   # [p+1]
        starts, cnt = bounds[:-1], bounds[1:] - bounds[:-1]
        rows = cnt * H
        rpad = (rows + (_TT_TILE - 1)) // _TT_TILE * _TT_TILE
        pend = torch.cumsum(rpad, 0)
        pbeg = pend - rpad
        self.ntiles = (n * H + _TT_TILE - 1) // _TT_TILE + p          # static upper bound
        self.mpad = self.ntiles * _TT_TILE
        t0 = torch.arange(self.ntiles, device=dev) * _TT_TILE
        g_t = torch.searchsorted(pend, t0, right=True)
        self.mtile_b = torch.where(g_t < p, g_t, torch.full_like(g_t, -1)).to(torch.int32)
        g_s = sorted_d.to(torch.int64)
        pos_sorted = pbeg[g_s] + (torch.arange(n, device=dev) - starts[g_s]) * H
        self.pos = torch.empty(n, dtype=torch.int64, device=dev)      # first row of every ORIGINAL lookup
        self.pos[perm] = pos_sorted
        # reduction segments of at most `seg` rows, none straddling a group
        chunks = (rows + (seg - 1)) // seg
        cend = torch.cumsum(chunks, 0)
        self.nseg = (n * H + seg - 1) // seg + p
        j = torch.arange(self.nseg, device=dev)
        g_j = torch.searchsorted(cend, j, right=True)
        live = g_j < p
        g_c = torch.where(live, g_j, torch.zeros_like(g_j))
        local = j - (cend[g_c] - chunks[g_c])
        k0 = pbeg[g_c] + local * seg
        K = torch.where(live, torch.minimum(rows[g_c] - local * seg, torch.full_like(local, seg)), torch.zeros_like(local))
        self.kseg = torch.stack([k0, K, g_c], 1).contiguous()          # int64 [nseg, 3]


def _move(src, src_row, src_stride, dst, dst_row, dst_stride, width, n, mask=None, accumulate=False):
    _lib.check(
        _lib.load().mi_move_chunks(src.data_ptr(), _lib.ptr(src_row), src_stride, dst.data_ptr(), _lib.ptr(dst_row),
                                   dst_stride, width, n, _lib.ptr(mask), 1 if accumulate else 0,
                                   _lib.stream_ptr(dst.device)),
        "mi_move_chunks",
    )


def tt_grouped_supported(n: int, q_shapes, ranks) -> bool:
    """The grouped form needs 16-byte chunk widths / strides, at least two cores, enough lookups to
    amortise the planning, and tile counts within one grid dimension."""
    ncores = len(q_shapes)
    if ncores < 2 or n < _TT_GROUPED_MIN:
        return False
    H = 1
    for c in range(ncores):
        H *= q_shapes[c]
        if (q_shapes[c] * ranks[c + 1]) % 4 or (c > 0 and ranks[c] % 4):
            return False
    return H % 4 == 0


def _tt_last_fast(H: int, K: int, q: int) -> bool:
    """The last level (rank 1 on the right) has a dedicated kernel when its H*q outputs tile a wave."""
    D = H * q
    return D <= 64 and 64 % D == 0 and K * q <= 512 and H * K <= 512


class TTLookupGrouped(torch.autograd.Function):
    """out[n, D] = TT-Rec rows: levels 1..n-2 as GEMMs grouped by that level's digit, the last level (q_last
    columns wide) by its own kernel straight from the previous level's layout."""

    @staticmethod
    def forward(ctx, idx, num_item: int, p_shapes, q_shapes, ranks, *cores):
        dev = _lib.require_gpu(idx, *cores)
        lib = _lib.load()
        stream = _lib.stream_ptr(dev)
        idxc = _i64c(idx).view(-1)
        cs = [_f32c(c) for c in cores]
        n, nc = idxc.numel(), len(cs)
        digits = torch.empty((nc, n), dtype=torch.int32, device=dev)
        valid = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(lib.mi_tt_digits(idxc.data_ptr(), n, num_item, (_ct.c_int32 * nc)(*p_shapes), nc, digits.data_ptr(),
                                    valid.data_ptr(), _lib.err_word(dev).data_ptr(), stream), "mi_tt_digits")
        Hs = [1]
        for q in q_shapes:
            Hs.append(Hs[-1] * q)                                   # Hs[c] = rows per lookup entering level c
        D = Hs[-1]
        last = nc - 1
        fast_last = _tt_last_fast(Hs[last], ranks[last], q_shapes[last])
        gemm_levels = list(range(1, last if fast_last else nc))
        plans = {c: _TTLevelPlan(digits[c], Hs[c], p_shapes[c], _TT_SEG) for c in gemm_levels}
        d0 = digits[0].to(torch.int64)
        w0 = q_shapes[0] * ranks[1]
        out = torch.empty((n, D), dtype=torch.float32, device=dev)
        saved_A = []
        src, src_row, src_stride = cs[0], d0, w0                    # where the next level finds a lookup's rows
        if gemm_levels:
            # level 1 operand: core 0's slices [q_0, r_1] dropped into level 1's layout
            A = torch.empty((plans[1].mpad, ranks[1]), dtype=torch.float32, device=dev)
            _move(cs[0], d0, w0, A, plans[1].pos, ranks[1], w0, n, mask=valid)
        for c in gemm_levels:
            K, Nn = ranks[c], q_shapes[c] * ranks[c + 1]
            C = torch.empty((plans[c].mpad, Nn), dtype=torch.float32, device=dev)
            _lib.check(lib.mi_gemm_f32_row_groups(A.data_ptr(), cs[c].data_ptr(), C.data_ptr(), plans[c].mpad, Nn, K,
                                                  K, Nn, Nn, 0, K * Nn, plans[c].mtile_b.data_ptr(), stream),
                       "mi_gemm_f32_row_groups")
            saved_A.append(A)
            width = Hs[c] * Nn                                      # a lookup's rows, contiguous
            if c + 1 in plans:
                A = torch.empty((plans[c + 1].mpad, ranks[c + 1]), dtype=torch.float32, device=dev)
                _move(C, plans[c].pos, Nn, A, plans[c + 1].pos, ranks[c + 1], width, n)
            elif c == last:
                _move(C, plans[c].pos, Nn, out, None, D, width, n)
            else:
                src, src_row, src_stride = C, plans[c].pos, Nn      # feeds the last level where it lies
        if fast_last:
            _lib.check(lib.mi_tt_last_fwd(src.data_ptr(), src_row.data_ptr(), src_stride, cs[last].data_ptr(),
                                          digits[last].data_ptr(), valid.data_ptr(), Hs[last], ranks[last],
                                          q_shapes[last], out.data_ptr(), n, stream), "mi_tt_last_fwd")
        ctx.save_for_backward(digits, valid, src if fast_last else digits, *cs, *saved_A)
        ctx.plans = plans
        ctx.meta = (num_item, list(p_shapes), list(q_shapes), list(ranks), Hs, n, fast_last)
        return out

    @staticmethod
    def backward(ctx, g):
        num_item, p_shapes, q_shapes, ranks, Hs, n, fast_last = ctx.meta
        nc = len(p_shapes)
        digits, valid, last_src, *rest = ctx.saved_tensors
        cs, As = rest[:nc], rest[nc:]
        plans = ctx.plans
        dev = g.device
        lib = _lib.load()
        stream = _lib.stream_ptr(dev)
        g = _f32c(g)
        D = Hs[-1]
        gcs = [torch.zeros_like(c) for c in cs]
        last = nc - 1
        d0 = digits[0].to(torch.int64)
        w0 = q_shapes[0] * ranks[1]
        plan0 = X = None

        def level0_buffer():
            # core 0's gradient: the lookups ordered by their first digit, each group's rows summed
            nonlocal plan0, X
            plan0 = _TTLevelPlan(digits[0], 1, p_shapes[0], _TT_SEG0)
            X = torch.empty((plan0.mpad, w0), dtype=torch.float32, device=dev)

        dC = None
        if fast_last:
            H, K, q = Hs[last], ranks[last], q_shapes[last]
            planL = _TTLevelPlan(digits[last], 1, p_shapes[last], 16)          # digit order, <= 16 lookups per wave
            order = torch.empty(planL.mpad, dtype=torch.int64, device=dev)
            order[planL.pos] = torch.arange(n, device=dev)
            if nc == 2:
                level0_buffer()
                src_row, src_stride, dst, dst_row, dst_stride = d0, w0, X, plan0.pos, w0
            else:
                Np = q_shapes[last - 1] * ranks[last]
                dC = torch.empty((plans[last - 1].mpad, Np), dtype=torch.float32, device=dev)
                src_row, src_stride, dst, dst_row, dst_stride = plans[last - 1].pos, Np, dC, plans[last - 1].pos, Np
            _lib.check(lib.mi_tt_last_bwd(last_src.data_ptr(), src_row.data_ptr(), src_stride, cs[last].data_ptr(),
                                          digits[last].data_ptr(), valid.data_ptr(), H, K, q, g.data_ptr(),
                                          dst.data_ptr(), dst_row.data_ptr(), dst_stride, order.data_ptr(),
                                          planL.kseg.data_ptr(), planL.nseg, gcs[last].data_ptr(), n, stream),
                       "mi_tt_last_bwd")
            top = last - 1
        else:
            N_last = q_shapes[last] * ranks[last + 1]
            dC = torch.empty((plans[last].mpad, N_last), dtype=torch.float32, device=dev)
            _move(g, None, D, dC, plans[last].pos, N_last, D, n, mask=valid)
            top = last
        for c in range(top, 0, -1):
            K, Nn = ranks[c], q_shapes[c] * ranks[c + 1]
            A = As[c - 1]
            # slice gradients: gcore_c[i] += A_rows(i)^T . dC_rows(i)
            _lib.check(lib.mi_gemm_f32_k_groups(A.data_ptr(), dC.data_ptr(), gcs[c].data_ptr(), K, Nn, K, Nn, Nn, K * Nn,
                                                plans[c].kseg.data_ptr(), plans[c].nseg, stream), "mi_gemm_f32_k_groups")
            # input gradient: dA = dC . core_c[i]^T
            dA = torch.empty((plans[c].mpad, K), dtype=torch.float32, device=dev)
            _lib.check(lib.mi_gemm_f32_row_groups(dC.data_ptr(), cs[c].data_ptr(), dA.data_ptr(), plans[c].mpad, K, Nn,
                                                  Nn, Nn, K, 1, K * Nn, plans[c].mtile_b.data_ptr(), stream),
                       "mi_gemm_f32_row_groups")
            width = Hs[c] * K
            if c > 1:
                Np = q_shapes[c - 1] * ranks[c]
                dC = torch.empty((plans[c - 1].mpad, Np), dtype=torch.float32, device=dev)
                _move(dA, plans[c].pos, K, dC, plans[c - 1].pos, Np, width, n)
            else:
                level0_buffer()
                _move(dA, plans[1].pos, K, X, plan0.pos, w0, w0, n, mask=valid)
        if X is not None:
            _lib.check(lib.mi_segment_sum(X.data_ptr(), w0, w0, plan0.kseg.data_ptr(), plan0.nseg,
                                          gcs[0].data_ptr(), w0, stream), "mi_segment_sum")
        return (None, None, None, None, None, *gcs)


def tt_lookup(idx, cores, num_item, p_shapes, q_shapes, ranks):
    args = (num_item, tuple(p_shapes), tuple(q_shapes), tuple(ranks), *cores)
    n = idx.numel()
    if not tt_grouped_supported(min(n, _TT_GROUPED_MIN), q_shapes, ranks) or n < _TT_GROUPED_MIN:
        return TTLookup.apply(idx, *args)
    # a level's row tiles must fit one grid dimension: very large batches (get_weight over the whole
    # vocabulary) go through in pieces
    H = 1
    for q in q_shapes[:-1]:
        H *= q
    piece = (_TT_MAX_TILES - max(p_shapes) - 1) * _TT_TILE // H
    flat = idx.reshape(-1)
    if n <= piece:
        return TTLookupGrouped.apply(flat, *args)
    return torch.cat([TTLookupGrouped.apply(flat[i:i + piece], *args) for i in range(0, n, piece)])


# --------------------------------------------------------------------------------------
# single-table gather with a per-element transform (PEP / RetrainPep)
class XformGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, W, S, M, srs: int, scs: int, xform: int):
        dev = _lib.require_gpu(idx, W)
        idxc, Wc = _i64c(idx), _f32c(W)
        Sc = None if S is None else _f32c(S)
        Mc = None if M is None else M.to(torch.uint8).contiguous()
        N, D = Wc.shape
        out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=dev)
        _lib.check(
            _lib.load().mi_xform_gather_fwd(idxc.data_ptr(), Wc.data_ptr(), _lib.ptr(Sc), _lib.ptr(Mc), srs, scs,
                                            out.data_ptr(), idxc.numel(), D, N, xform,
                                            _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
            "mi_xform_gather_fwd",
        )
        ctx.save_for_backward(idxc, Wc, Sc, Mc)
        ctx.meta = (srs, scs, xform, tuple(W.shape), None if S is None else tuple(S.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        idxc, Wc, Sc, Mc = ctx.saved_tensors
        srs, scs, xform, Wshape, Sshape = ctx.meta
        _no_atomics_promised("the transformed (PEP / masked) gather backward")
        g = _f32c(g)
        N, D = Wc.shape
        gW = torch.zeros_like(Wc)
        gS = torch.zeros_like(Sc) if (Sc is not None and xform == XF_SOFT) else None
        _lib.check(
            _lib.load().mi_xform_gather_bwd(idxc.data_ptr(), g.data_ptr(), Wc.data_ptr(), _lib.ptr(Sc), _lib.ptr(Mc),
                                            srs, scs, gW.data_ptr(), _lib.ptr(gS), gS.numel() if gS is not None else 0,
                                            idxc.numel(), D, N, xform, _lib.stream_ptr(g.device)),
            "mi_xform_gather_bwd",
        )
        return None, gW.view(Wshape), (gS.view(Sshape) if gS is not None else None), None, None, None, None


def soft_threshold_gather(idx, W, s):
    """F.embedding(idx, sign(W)*relu(|W| - sigmoid(s))) with s broadcastable to W ([1], [D], [N,1], [N,D])."""
    N, D = W.shape
    if s.numel() == 1:
        srs, scs = 0, 0
    elif s.dim() == 1 and s.shape[0] == D:
        srs, scs = 0, 1
    elif tuple(s.shape) == (N, 1):
        srs, scs = 1, 0
    elif tuple(s.shape) == (N, D):
        srs, scs = D, 1
    else:
        raise ValueError(f"threshold of shape {tuple(s.shape)} does not broadcast over a [{N},{D}] table")
    return XformGather.apply(idx, W, s, None, srs, scs, XF_SOFT)


def masked_gather(idx, W, mask):
    return XformGather.apply(idx, W, None, mask, 0, 0, XF_MASK)


class _MaskedGatherRowGrad(torch.autograd.Function):
    """out = (W * mask)[idx] with W.grad in ROW FORM (uncoalesced COO over idx, values g * mask[idx]) — what the reference's
    F.embedding(x, weight * mask, sparse=True) hands torch.optim.SparseAdam (src/models/embeddings/pep_embedding.py:215-221)."""

    @staticmethod
    def forward(ctx, idx, W, mask):
        with torch.no_grad():
            out = masked_gather(idx, W, mask)
        ctx.save_for_backward(_i64c(idx).reshape(-1), mask)
        ctx.wshape = tuple(W.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        flat, mask = ctx.saved_tensors
        D = ctx.wshape[1]
        vals = _f32c(g).reshape(-1, D) * mask.index_select(0, flat).to(torch.float32)
        return None, _coo(flat, vals, ctx.wshape), None


def masked_gather_row_grad(idx, W, mask):
    return _MaskedGatherRowGrad.apply(idx, W, mask)


class _DualMaskedGatherRowGrad(torch.autograd.Function):
    """out = (T1 * M1)[idx % mod1] + (T2 * M2)[idx // div2] with both gradients in row form (RetrainCerpEmbedding with
    sparse=True, src/models/embeddings/cerp_embedding.py:329-367)."""

    @staticmethod
    def forward(ctx, idx, T1, T2, M1, M2, mod1: int, div2: int):
        with torch.no_grad():
            out = dual_gather(idx, T1, T2, mod1=mod1, div2=div2, op="add", M1=M1, M2=M2)
        flat = _i64c(idx).reshape(-1)
        ctx.save_for_backward(flat, M1, M2)
        ctx.meta = (int(mod1), int(div2), tuple(T1.shape), tuple(T2.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        flat, M1, M2 = ctx.saved_tensors
        mod1, div2, s1, s2 = ctx.meta
        g2 = _f32c(g).reshape(-1, s1[1])
        i1, i2 = flat % mod1, torch.div(flat, div2, rounding_mode="floor")
        g1 = _coo(i1, g2 * M1.index_select(0, i1).to(torch.float32), s1) if ctx.needs_input_grad[1] else None
        gq = _coo(i2, g2 * M2.index_select(0, i2).to(torch.float32), s2) if ctx.needs_input_grad[2] else None
        return None, g1, gq, None, None, None, None


def dual_masked_gather_row_grad(idx, T1, T2, M1, M2, mod1: int, div2: int):
    return _DualMaskedGatherRowGrad.apply(idx, T1, T2, M1, M2, mod1, div2)


def gather_rows_quant(idx, W, scale=None, bias=None):
    """Dequantising row gather for fp16 / int8 / int16 tables (inference only)."""
    dev = _lib.require_gpu(idx, W)
    qtype = {torch.float16: 1, torch.int8: 2, torch.int16: 3}.get(W.dtype)
    if qtype is None:
        raise TypeError(f"unsupported quantised table dtype {W.dtype}")
    idxc, Wc = _i64c(idx), W.contiguous()
    N, D = Wc.shape
    out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=dev)
    _lib.check(
        _lib.load().mi_gather_rows_quant(idxc.data_ptr(), Wc.data_ptr(), qtype, _lib.ptr(scale), _lib.ptr(bias),
                                         out.data_ptr(), idxc.numel(), D, N, _lib.err_word(dev).data_ptr(),
                                         _lib.stream_ptr(dev)),
        "mi_gather_rows_quant",
    )
    return out


# ---- row-sharded lookup: device-side routing (csrc/route.hip) -----------------
def route_buckets(idx: torch.Tensor, offsets: Optional[torch.Tensor], world: int, num_rows: int, cap: int,
                  overflow: torch.Tensor, slot_out: Optional[torch.Tensor] = None):
    """Stable bucketing of rows = idx + offsets by owner = row % world.

    Returns (send_rows int64[world*cap] of owner-local rows, slot int64[idx.shape]); `overflow`
    is an int32[1] device word that gets 1 OR-ed in when a bucket exceeds `cap`."""
    dev = _lib.require_gpu(idx, offsets, overflow)
    lib = _lib.load()
    idxc = _i64c(idx)
    if overflow.dtype != torch.int32:
        raise TypeError("overflow must be an int32 device word")
    F = idxc.shape[-1] if idxc.dim() > 1 else 1
    off = None if offsets is None else _i64c(offsets.reshape(-1))
    if off is not None and off.numel() != F:
        raise ValueError(f"offsets must have {F} entries")
    n = idxc.numel()
    ws = torch.empty(int(lib.mi_route_workspace_elems(n, world)), dtype=torch.int32, device=dev)
    send_rows = torch.empty(world * cap, dtype=torch.int64, device=dev)
    slot = slot_out if slot_out is not None else torch.empty(idxc.shape, dtype=torch.int64, device=dev)
    if slot.numel() != n or slot.dtype != torch.int64 or not slot.is_contiguous():
        raise ValueError("slot_out must be a contiguous int64 tensor with one entry per lookup")
    _lib.check(
        lib.mi_route_buckets(idxc.data_ptr(), _lib.ptr(off), n, F, world, num_rows, cap, ws.data_ptr(),
                             send_rows.data_ptr(), slot.data_ptr(), overflow.data_ptr(),
                             _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
        "mi_route_buckets",
    )
    return send_rows, slot


def gather_pack_rows(local_rows: torch.Tensor, W: torch.Tensor, w1: torch.Tensor) -> torch.Tensor:
    """[m, D+4] packed rows (W[r,:], w1[r], 0, 0, 0) for the owner-local rows r."""
    dev = _lib.require_gpu(local_rows, W, w1)
    lib = _lib.load()
    rows = _i64c(local_rows.reshape(-1))
    Wc, w1c = _f32c(W), _f32c(w1)
    Nl, D = Wc.shape
    if w1c.numel() != Nl:
        raise ValueError("first-order shard must have one weight per embedding row")
    out = torch.empty((rows.numel(), D + 4), dtype=torch.float32, device=dev)
    _lib.check(
        lib.mi_gather_pack_rows(rows.data_ptr(), Wc.data_ptr(), w1c.data_ptr(), out.data_ptr(), rows.numel(), D,
                                Nl, _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
        "mi_gather_pack_rows",
    )
    return out


def unpack_rows(packed: torch.Tensor, D: int):
    """([m, D], [m]) contiguous copies of columns 0..D-1 and column D of packed rows [m, D+4] — one launch (mi_unpack_rows)."""
    dev = _lib.require_gpu(packed)
    pc = _f32c(packed)
    if pc.dim() != 2 or pc.shape[1] != D + 4:
        raise ValueError("unpack_rows: rows of D + 4 floats expected")
    m = pc.shape[0]
    vals = torch.empty((m, D), dtype=torch.float32, device=dev)
    lin = torch.empty((m,), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().mi_unpack_rows(pc.data_ptr(), vals.data_ptr(), lin.data_ptr(), m, D, _lib.stream_ptr(dev)),
               "mi_unpack_rows")
    return vals, lin


class SlotFM(torch.autograd.Function):
    """(emb[B,F,D], y_fm[B]) from the packed rows a sharded lookup received, addressed by slot
    (src/models/deepfm.py:88-98 on exchanged rows).  buf is [S+1, D+4] with row S all zeros (the
    dump slot); the gradient w.r.t. buf comes back in the same layout, ready for the all-to-all."""

    @staticmethod
    def forward(ctx, buf, slot, bias):
        dev = _lib.require_gpu(buf, slot, bias)
        lib = _lib.load()
        buf = _f32c(buf)
        slot = _i64c(slot)
        B, F = slot.shape
        D = buf.shape[1] - 4
        emb = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        yfm = torch.empty((B,), dtype=torch.float32, device=dev)
        _lib.check(
            lib.mi_slot_fm_fwd(slot.data_ptr(), buf.data_ptr(), buf.shape[0], _lib.ptr(bias), emb.data_ptr(),
                               yfm.data_ptr(), B, F, D, _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)),
            "mi_slot_fm_fwd",
        )
        ctx.save_for_backward(emb, slot)
        ctx.meta = (B, F, D, buf.shape[0], bias is not None)
        ctx.set_materialize_grads(False)
        return emb, yfm

    @staticmethod
    def backward(ctx, g_emb, g_y):
        emb, slot = ctx.saved_tensors
        B, F, D, rows, has_bias = ctx.meta
        dev = emb.device
        lib = _lib.load()
        if g_y is None:
            g_y = torch.zeros((B,), dtype=torch.float32, device=dev)
        g_y = _f32c(g_y)
        g_emb = None if g_emb is None else _f32c(g_emb)
        gbuf = None
        want_gb = has_bias and ctx.needs_input_grad[2]
        gb = torch.empty((1,), dtype=torch.float32, device=dev) if want_gb else None
        if ctx.needs_input_grad[0]:
            gbuf = torch.empty((rows, D + 4), dtype=torch.float32, device=dev)
            # the kernel zeroes rows [0, S) itself; the dump row's gradient is never read
            _lib.check(
                lib.mi_slot_fm_bwd(slot.data_ptr(), emb.data_ptr(), g_y.data_ptr(), _lib.ptr(g_emb),
                                   gbuf.data_ptr(), _lib.ptr(gb), rows - 1, B, F, D, _lib.stream_ptr(dev)),
                "mi_slot_fm_bwd",
            )
        elif want_gb:
            gb = g_y.sum().view(1)
        return gbuf, None, gb


def slot_fm(buf, slot, bias):
    return SlotFM.apply(buf, slot, bias)
