"""OptEmbed supernet embedding for DeepFM — reference: src/models/embeddings/deepfm_opt_embed.py:39-330 and
src/models/embeddings/optembed_utils.py:10-112.

Drop-in for the LOOKUP side of the class: same constructor, parameters (`_weight`, `_mask_e_module._t_param`),
buffers (`_full_mask_d`, `_mask_e_module._field_dims`), `forward` (training: a fresh uniform dimension mask per
(sample, field) and the per-field row mask; eval: lookups of the masked table), `get_weight(mask_d)`, `get_l_s`,
`get_sparsity`, `get_num_params`, `get_mask_e`, `get_submask`.  Gather, L1/L2 row norm, BinaryStep row mask and
triangular dimension mask are ONE HIP kernel (mi_optembed_fwd); its backward carries BinaryStep's surrogate
gradient to the table and the thresholds.  The evolutionary-search helpers of the reference file
(_generate_candidate, _crossover, _mutate, evol_search_deepfm) are host-side search scripts outside the hot path
and are not mirrored; the eval path computes masked rows on the fly instead of caching a masked copy of the table.
"""
from typing import List, Optional, Union

import torch
from torch import nn

from .. import _kernels, _lib
from .base import IEmbedding


def get_mask(hidden_size: int) -> torch.Tensor:
    """matrix[i][j] = 1 if i >= j (optembed_utils.py:10-22)."""
    return torch.tril(torch.ones((hidden_size, hidden_size), dtype=torch.bool))


class _OptLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, t, idx, tix, F: int, dmax, norm: int):
        dev = _lib.require_gpu(W, idx)
        Wc = _kernels._f32c(W)
        idxc = _kernels._i64c(idx)
        tc = None if t is None else _kernels._f32c(t)
        tixc = None if tix is None else _kernels._i64c(tix)
        dmc = None if dmax is None else _kernels._i64c(dmax)
        N, D = Wc.shape
        out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().mi_optembed_fwd(idxc.data_ptr(), Wc.data_ptr(), _lib.ptr(tc), _lib.ptr(tixc), F,
                                               _lib.ptr(dmc), norm, out.data_ptr(), idxc.numel(), D, N,
                                               _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)), "mi_optembed_fwd")
        ctx.save_for_backward(Wc, tc, idxc, tixc, dmc)
        ctx.meta = (F, norm, tuple(W.shape), None if t is None else tuple(t.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        Wc, tc, idxc, tixc, dmc = ctx.saved_tensors
        F, norm, Wshape, tshape = ctx.meta
        g = _kernels._f32c(g)
        N, D = Wc.shape
        dW = torch.zeros_like(Wc) if ctx.needs_input_grad[0] else None
        dt = torch.zeros_like(tc) if (tc is not None and ctx.needs_input_grad[1]) else None
        _lib.check(_lib.load().mi_optembed_bwd(idxc.data_ptr(), Wc.data_ptr(), _lib.ptr(tc), _lib.ptr(tixc), F,
                                               _lib.ptr(dmc), norm, g.data_ptr(), _lib.ptr(dW), _lib.ptr(dt),
                                               idxc.numel(), D, N, _lib.stream_ptr(g.device)), "mi_optembed_bwd")
        return (None if dW is None else dW.view(Wshape), None if dt is None else dt.view(tshape), None, None, None,
                None, None)


class _MaskEmbeddingModule(nn.Module):
    """Holder of the thresholds (optembed_utils.py:46-86); the masking itself runs inside the lookup kernel."""

    def __init__(self, field_dims: torch.Tensor, t_init: float = 0, mode_threshold_e="field", norm=1):
        super().__init__()
        assert mode_threshold_e in ["feature", "field"]
        self.mode_threshold_e = mode_threshold_e
        self.register_buffer("_field_dims", field_dims)
        self._num_item = int(field_dims.sum())
        self._num_field = len(field_dims)
        self._t_param = nn.Parameter(torch.full((self._num_item if mode_threshold_e == "feature" else self._num_field,),
                                                float(t_init)))
        self._norm = norm


class OptEmbed(IEmbedding):
    def __init__(self, field_dims: Union[List[int], int], hidden_size: int, mode: Optional[str] = None,
                 t_init: Optional[float] = 0, mode_threshold_e="field", mode_threshold_d="field", norm=1,
                 target_sparsity: Optional[float] = None):
        super().__init__()
        if isinstance(field_dims, int):
            field_dims = [field_dims]
        assert mode in ["sum", "mean", "max", None]
        assert mode_threshold_e in ["field", "feature"]
        assert mode_threshold_d in ["field", "feature"]
        self._field_dims = torch.tensor(field_dims, dtype=torch.int64)
        self._num_item = int(self._field_dims.sum())
        self._num_field = len(field_dims)
        self._hidden_size = hidden_size
        self._weight = nn.Parameter(torch.empty((self._num_item, hidden_size)))
        nn.init.xavier_uniform_(self._weight)
        self._mode = mode
        self._t_init = t_init
        self._norm = norm
        self._mask_e_module = (nn.Identity() if t_init is None
                               else _MaskEmbeddingModule(self._field_dims, t_init, mode_threshold_e, norm))
        self.register_buffer("_full_mask_d", get_mask(hidden_size))
        # field of every row: thresholds / dimension masks given per field are addressed through it
        self.register_buffer("_row_field", torch.repeat_interleave(torch.arange(self._num_field), self._field_dims),
                             persistent=False)
        self._target_sparsity = target_sparsity
        self._mode_d = mode_threshold_d
        self._eval_mask_d = None           # set by get_weight(mask_d) for the eval lookups that follow

    # ---- pieces of the kernel call ---------------------------------------------------------------
    def _thresholds(self):
        return None if self._t_init is None else self._mask_e_module._t_param

    def _per_row(self, values: torch.Tensor, rows: torch.Tensor, per_field: bool) -> torch.Tensor:
        """values given per field or per feature -> one per looked-up row."""
        return values[self._row_field[rows]] if per_field else values[rows]

    def _lookup(self, rows, mask_d):
        t = self._thresholds()
        tix = None
        if t is not None:
            field_t = self._mask_e_module.mode_threshold_e == "field"
            tix = self._row_field[rows] if field_t else rows
        dmax = None
        if mask_d is not None:
            if mask_d.dtype == torch.bool:         # an explicit [num_item, D] mask: its row sums - 1 (prefix masks)
                dmax = self._per_row(mask_d.sum(1) - 1, rows, False)
            else:
                dmax = self._per_row(mask_d.to(rows.device), rows, self._mode_d == "field")
        return _OptLookup.apply(self._weight, t, rows, tix, self._num_field, dmax, self._norm)

    # ---- reference API -----------------------------------------------------------------------------
    def get_l_s(self):
        if self._t_init is None:
            return 0
        return torch.exp(-self._mask_e_module._t_param).sum()

    def get_weight(self, mask_d: Optional[torch.Tensor] = None):
        """The masked table (deepfm_opt_embed.py:148-200).  Training with mask_d=None samples a uniform dimension
        mask (per field or per feature, `mode_threshold_d`); eval with mask_d=None applies the row mask only."""
        dev = self._weight.device
        if self.training and mask_d is None:
            size = self._num_field if self._mode_d == "field" else self._num_item
            hidden = self._hidden_size
            if self._target_sparsity is not None and self._mode_d == "feature":
                assert self._target_sparsity >= 0.5, "Generate naive only could generate sparsity from 0.5"
                hidden = int(hidden * 2 * (1 - self._target_sparsity))
            mask_d = torch.randint(0, hidden, (size,), device=dev)
        if not self.training:
            self._eval_mask_d = mask_d
        return self._lookup(torch.arange(self._num_item, device=dev), mask_d)

    def forward(self, x, mask_d=None):
        """x: rows after offsets, [B, num_field] (or any shape in eval).  Training: every (sample, field) draws its own
        number of kept dimensions (deepfm_opt_embed.py:219-225; the argument is ignored as in the reference);
        `_forced_mask_d` (tests) replaces that draw."""
        if self.training:
            forced = self.__dict__.get("_forced_mask_d")
            dmax = forced if forced is not None else torch.randint(0, self._hidden_size, size=tuple(x.shape),
                                                                    device=self._weight.device)
            t = self._thresholds()
            if t is not None:
                assert self._mask_e_module.mode_threshold_e == "field", "Cannot apply field mask to input"
            emb = _OptLookup.apply(self._weight, t, x, None, self._num_field, dmax, self._norm)
        else:
            if mask_d is not None:
                self._eval_mask_d = mask_d
            emb = self._lookup(x, self._eval_mask_d)
        return _kernels.bag_reduce(emb, self._mode)

    def get_sparsity(self, get_n_params=False):
        with torch.no_grad():
            training, self.training = self.training, False
            emb = self._lookup(torch.arange(self._num_item, device=self._weight.device), None)
            self.training = training
        nnz = int(torch.count_nonzero(emb).item())
        sparsity = 1 - nnz / (emb.shape[0] * emb.shape[1])
        return (sparsity, nnz) if get_n_params else sparsity

    def get_num_params(self):
        return self.get_sparsity(True)[1]

    def get_mask_e(self):
        if self._t_init is None:
            return torch.ones(self._num_item, dtype=int)
        with torch.no_grad():
            emb = self._lookup(torch.arange(self._num_item, device=self._weight.device), None)
        return (emb.norm(1, 1) > 0).to(int).cpu()

    def get_submask(self) -> torch.Tensor:
        """Features still alive per unit of the dimension mask (per field or per feature)."""
        mask_e = self.get_mask_e()
        if self._mode_d == "feature":
            return mask_e
        out = torch.zeros(self._num_field, dtype=mask_e.dtype)
        return out.index_add_(0, self._row_field.cpu(), mask_e)
