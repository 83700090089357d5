"""Quantisation-aware-training embedding — reference: src/models/embeddings/qat_emb.py:87-122.

Drop-in: a VanillaEmbedding (same `_emb_module.weight` key) plus the learnable scalar `scale`
(initialised (max - min) / (q_max - q_min) over the table, frozen with fixed_scale=True); `forward` is the
row gather followed by StotasticRounding, `get_weight` the raw table.  Gather, clamp, stochastic rounding
and rescale are ONE HIP kernel (mi_qat_gather_fwd); the backward (straight-through to the rows, the
reference's scale gradient) re-derives the rounding from the counter generator instead of storing it.
"""
from typing import List, Union

import torch
from torch import nn

from .. import _kernels, _lib
from .base import VanillaEmbedding


def get_qmax_qmin(n_bits):
    n = int(n_bits)
    return (1 << (n - 1)) - 1, -(1 << (n - 1))


class _QatLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, W, idx, n_bits: int, seed, salt: int, prob, dense_shape):
        dev = _lib.require_gpu(W, scale)
        Wc = _kernels._f32c(W)
        idxc = None if idx is None else _kernels._i64c(idx)
        N, D = Wc.shape
        n = N if idxc is None else idxc.numel()
        lead = (N,) if idxc is None else tuple(idx.shape)
        out = torch.empty(lead + (D,), dtype=torch.float32, device=dev)
        probc = None if prob is None else _kernels._f32c(prob)
        sc = _kernels._f32c(scale.reshape(1))
        _lib.check(_lib.load().mi_qat_gather_fwd(_lib.ptr(idxc), Wc.data_ptr(), sc.data_ptr(), n_bits, _lib.ptr(probc),
                                                 _lib.ptr(seed), salt, out.data_ptr(), n, D, N,
                                                 _lib.err_word(dev).data_ptr(), _lib.stream_ptr(dev)), "mi_qat_gather_fwd")
        ctx.save_for_backward(Wc, sc, idxc, seed, probc)
        ctx.meta = (n_bits, salt, n, D, N, tuple(W.shape), tuple(scale.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        Wc, sc, idxc, seed, probc = ctx.saved_tensors
        n_bits, salt, n, D, N, Wshape, sshape = ctx.meta
        g = _kernels._f32c(g)
        dW = torch.zeros_like(Wc) if ctx.needs_input_grad[1] else None
        ds = torch.zeros(1, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.load().mi_qat_gather_bwd(_lib.ptr(idxc), Wc.data_ptr(), sc.data_ptr(), n_bits, _lib.ptr(probc),
                                                 _lib.ptr(seed), salt, g.data_ptr(), _lib.ptr(dW), _lib.ptr(ds), n, D, N,
                                                 _lib.stream_ptr(g.device)), "mi_qat_gather_bwd")
        return (None if ds is None else ds.view(sshape), None if dW is None else dW.view(Wshape),
                None, None, None, None, None, None)


class QAT_EmbInt(VanillaEmbedding):
    def __init__(self, field_dims: Union[int, List[int]], num_factor: int = 16, mode=None, initializer="xavier", *,
                 stochastic_rounding: bool = True, n_bits=8, fixed_scale=False, **kwargs):
        super().__init__(field_dims, num_factor, mode, initializer, **kwargs)
        assert n_bits in [8, 16]
        assert stochastic_rounding, "Not implement deterministic yet, I am lazy, please wait"
        self.n_bits = torch.tensor(n_bits)
        q_max, q_min = get_qmax_qmin(n_bits)
        with torch.no_grad():
            w = super().get_weight()
            scale_init = (w.max() - w.min()) / (q_max - q_min)
        self.register_parameter("scale", nn.Parameter(scale_init, not fixed_scale))
        # the rounding stream: a device word advanced once per forward (capture-safe), snapshotted for the backward
        self.register_buffer("_sr_seed", torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFF], dtype=torch.int64),
                             persistent=False)

    def forward(self, x, prob=None):
        """prob (optional, tests): the uniform draw to round with, shaped like the output."""
        seed = None
        if prob is None:
            seed = self._sr_seed.clone()
            self._sr_seed.add_(1)
        W = self._emb_module.weight
        if self._mode is None:
            return _QatLookup.apply(self.scale, W, x, int(self.n_bits), seed, 0, prob, None)
        bag = super().forward(x)                       # EmbeddingBag modes reduce first, then the bag is rounded
        return _QatLookup.apply(self.scale, bag, None, int(self.n_bits), seed, 0, prob, None)

    def get_weight(self):
        return super().get_weight()
