"""DHEmbedding's MLP (reference src/models/embeddings/dh_embedding.py:100-117,345-356) on the library's
own kernels: per layer Linear -> Mish (use_bn 0) | Linear -> BatchNorm1d -> Mish (use_bn 2, the default) | Linear -> Mish ->
BatchNorm1d (use_bn 1, the LightGCN configs), training (batch statistics) and eval (running statistics), one autograd node.

    contractions   z = x W^T on the MFMA product of csrc/tail_gemm.hpp (mi_tail_fwd_gemm_s, plain operand: the first layer's
                   k = 1024 hash features are the dominant cost); dx = dz W on mi_tail_dgrad_gemm_s (dz = the BatchNorm
                   backward applied in its operand load); dW = dz^T x in one multi-problem launch (mi_gemm_f32_multi)
    element work   csrc/mish_mlp.hip: column-wise affine + Mish forward / backward with the BatchNorm statistics and the
                   dgamma / dbeta column sums as per-tile partials, joined by the tail's finalize kernels
                   (mi_tail_bn_finalize_fwd updates the running statistics exactly as F.batch_norm(training=True))

`mish_mlp_plan(seq, use_bn, x)` decides whether the Sequential matches (fp32, widths multiples of 4, BatchNorm1d with affine
parameters and running statistics); everything else keeps nn.Sequential."""
import ctypes
from typing import List, Optional

import torch
from torch import nn

from . import _kernels, _lib

_ONES = {}


def _ones(n: int, dev) -> torch.Tensor:
    key = (n, str(dev))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((n,), dtype=torch.float32, device=dev)
    return t


class _L:
    __slots__ = ("lin", "bn")

    def __init__(self, lin, bn):
        self.lin, self.bn = lin, bn


def mish_mlp_plan(seq: nn.Sequential, use_bn: int, x: torch.Tensor) -> Optional[List[_L]]:
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 1):
        return None
    mods, plan, i, width = list(seq), [], 0, x.shape[1]
    per = 3 if use_bn in (1, 2) else 2
    if len(mods) == 0 or len(mods) % per:
        return None
    while i < len(mods):
        grp = mods[i:i + per]
        lin = grp[0]
        if use_bn == 1:
            ok = isinstance(grp[1], nn.Mish) and isinstance(grp[2], nn.BatchNorm1d)
            bn = grp[2] if ok else None
        elif use_bn == 2:
            ok = isinstance(grp[1], nn.BatchNorm1d) and isinstance(grp[2], nn.Mish)
            bn = grp[1] if ok else None
        else:
            ok, bn = isinstance(grp[1], nn.Mish), None
        if not (ok and isinstance(lin, nn.Linear) and lin.in_features == width and width % 4 == 0 and lin.out_features % 4 == 0):
            return None
        if bn is not None and (not bn.affine or not bn.track_running_stats or bn.momentum is None):
            return None
        if bn is not None and bn.training and x.shape[0] < 2:
            return None
        ts = (lin.weight, lin.bias) + ((bn.weight, bn.bias, bn.running_mean, bn.running_var) if bn is not None else ())
        if any(t is not None and (t.dtype != torch.float32 or t.device != x.device) for t in ts):
            return None
        plan.append(_L(lin, bn))
        width = lin.out_features
        i += per
    return plan


def _product(x, W, Z, part, M, N, K, s):
    _lib.check(_lib.load().mi_tail_fwd_gemm_s(x.data_ptr(), K, None, None, None, 0.0, None, W.data_ptr(), K, Z.data_ptr(), N,
                                              _lib.ptr(part), None, M, N, K, None, 0, None, None, None, s), "mi_tail_fwd_gemm_s")


def _finalize_fwd(part, M, N, L, mean_offset, c, s):
    bn = L.bn
    _lib.check(_lib.load().mi_tail_bn_finalize_fwd(
        part.data_ptr(), M, N, bn.weight.data_ptr(), bn.bias.data_ptr(), _lib.ptr(mean_offset), bn.running_mean.data_ptr(),
        bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps), bn.num_batches_tracked.data_ptr(), None,
        c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), s), "mi_tail_bn_finalize_fwd")


def _eval_consts(L, with_bias: bool, c, s):
    """(mu, sc, be, rstd) of an eval-mode BatchNorm1d: mu = running_mean (- the Linear bias when it sits in front), ..."""
    P = ctypes.c_void_p * 1
    bn = L.bn
    widths, eps = (ctypes.c_int32 * 1)(c.shape[1]), (ctypes.c_float * 1)(float(bn.eps))
    gam, bet, rme, rva = P(bn.weight.data_ptr()), P(bn.bias.data_ptr()), P(bn.running_mean.data_ptr()), P(bn.running_var.data_ptr())
    bia = P((_lib.ptr(L.lin.bias) or 0) if with_bias else 0)
    outs = [P(c[r].data_ptr()) for r in range(4)]
    _lib.check(_lib.load().mi_tail_affine_consts(
        1, ctypes.addressof(widths), ctypes.addressof(gam), ctypes.addressof(bet), ctypes.addressof(rme), ctypes.addressof(rva),
        ctypes.addressof(bia), ctypes.addressof(eps), ctypes.addressof(outs[0]), ctypes.addressof(outs[1]),
        ctypes.addressof(outs[2]), ctypes.addressof(outs[3]), s), "mi_tail_affine_consts")


class MishMlpFn(torch.autograd.Function):
    """out = seq(x) for DHEmbedding's layer pattern.  Differentiable inputs: x, then per layer (W, b, gamma, beta)
    (gamma / beta None without BatchNorm)."""

    @staticmethod
    def forward(ctx, plan, use_bn: int, grad: bool, x, *params):
        lib = _lib.load()
        dev = _lib.require_gpu(x)
        s = _lib.stream_ptr(dev)
        x = _kernels._f32c(x)
        M = x.shape[0]
        saved, meta = [], []
        a = x
        for i, L in enumerate(plan):
            W = _kernels._f32c(params[4 * i])
            b = L.lin.bias
            N, K = W.shape
            training = L.bn is not None and L.bn.training
            Z = torch.empty((M, N), dtype=torch.float32, device=dev)
            c = torch.empty((4, N), dtype=torch.float32, device=dev) if L.bn is not None else None
            m_act = None
            out = torch.empty((M, N), dtype=torch.float32, device=dev)
            if use_bn == 2:
                part = torch.empty(int(lib.mi_tail_part_elems(M, N)), dtype=torch.float32, device=dev) if training else None
                _product(a, W, Z, part, M, N, K, s)
                if training:                                  # statistics of z; the bias only shifts the running mean
                    _finalize_fwd(part, M, N, L, b, c, s)
                else:
                    _eval_consts(L, True, c, s)
                _lib.check(lib.mi_col_act_fwd(Z.data_ptr(), N, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), 1, out.data_ptr(),
                                              None, M, N, s), "mi_col_act_fwd")
            elif use_bn == 1:
                _product(a, W, Z, None, M, N, K, s)
                m_act = torch.empty((M, N), dtype=torch.float32, device=dev)
                part = torch.empty(int(lib.mi_tail_part_elems(M, N)), dtype=torch.float32, device=dev) if training else None
                _lib.check(lib.mi_col_act_fwd(Z.data_ptr(), N, None, None, _lib.ptr(b), 1, m_act.data_ptr(), _lib.ptr(part), M, N, s),
                           "mi_col_act_fwd")
                if training:                                  # statistics of mish(z + b) itself
                    _finalize_fwd(part, M, N, L, None, c, s)
                else:
                    _eval_consts(L, False, c, s)
                _lib.check(lib.mi_col_act_fwd(m_act.data_ptr(), N, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), 0, out.data_ptr(),
                                              None, M, N, s), "mi_col_act_fwd")
            else:
                _product(a, W, Z, None, M, N, K, s)
                _lib.check(lib.mi_col_act_fwd(Z.data_ptr(), N, None, None, _lib.ptr(b), 1, out.data_ptr(), None, M, N, s), "mi_col_act_fwd")
            if grad:
                e = x.new_empty(0)
                saved += [a, W, Z, c if c is not None else e, m_act if m_act is not None else e,
                          b if b is not None else e, L.bn.weight if L.bn is not None else e]
                meta.append(training)
            a = out
        ctx.plan, ctx.use_bn, ctx.meta = plan, use_bn, meta
        ctx.save_for_backward(*saved)
        return a

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        plan, use_bn = ctx.plan, ctx.use_bn
        saved = ctx.saved_tensors
        need = ctx.needs_input_grad            # (plan, use_bn, grad, x, *params)
        dev = g.device
        s = _lib.stream_ptr(dev)
        g = _kernels._f32c(g)
        M = g.shape[0]
        MT = (M + 63) // 64
        grads: List[Optional[torch.Tensor]] = [None] * (4 * len(plan))
        later = []
        for i in range(len(plan) - 1, -1, -1):
            a_in, W, Z, c, m_act, b, gamma = saved[7 * i: 7 * i + 7]
            c = c if c.numel() else None
            m_act = m_act if m_act.numel() else None
            b = b if b.numel() else None
            gamma = gamma if gamma.numel() else None
            training = ctx.meta[i]
            N, K = W.shape
            need_dx = i > 0 or need[3]
            need_W, need_b = need[4 + 4 * i], need[4 + 4 * i + 1] and b is not None
            part = torch.empty((MT, N, 2), dtype=torch.float32, device=dev)
            dzc = torch.empty((3, N), dtype=torch.float32, device=dev)       # al, bz, de
            dgb = torch.empty((2, N), dtype=torch.float32, device=dev) if c is not None else None
            dbias = torch.empty((N,), dtype=torch.float32, device=dev) if (need_b and not (use_bn == 2 and training)) else None
            ones = _ones(N, dev)

            def finalize(p, affine, want_dbias, with_bn):
                _lib.check(lib.mi_tail_bn_finalize_bwd_a(
                    p.data_ptr(), MT, M, N, _lib.ptr(gamma) if with_bn else None, (c[3] if with_bn else ones).data_ptr(),
                    dgb[0].data_ptr() if with_bn else None, dgb[1].data_ptr() if with_bn else None, dzc[0].data_ptr(),
                    dzc[1].data_ptr(), dzc[2].data_ptr(), None, 0, None, None, int(affine), _lib.ptr(dbias) if want_dbias else None, s),
                    "mi_tail_bn_finalize_bwd_a")

            dz_plain = None               # dz as a matrix (plain operand of the products), or None: (dy, z, constants)
            dy = None
            if use_bn == 1:
                # BatchNorm backward constants from (sum g, sum g (m - mu)), then dz = dm * mish'(z + b) and its column sums
                _lib.check(lib.mi_col_act_bwd(g.data_ptr(), m_act.data_ptr(), N, c[0].data_ptr(), None, None, 0, None, part.data_ptr(),
                                              M, N, s), "mi_col_act_bwd")
                finalize(part, not training, False, True)
                dz_plain = torch.empty((M, N), dtype=torch.float32, device=dev)
                part2 = torch.empty((MT, N, 2), dtype=torch.float32, device=dev)
                _lib.check(lib.mi_bn_mish_bwd(g.data_ptr(), m_act.data_ptr(), Z.data_ptr(), N, _lib.ptr(b), c[0].data_ptr(),
                                              dzc[0].data_ptr(), dzc[1].data_ptr(), dzc[2].data_ptr(), dz_plain.data_ptr(),
                                              part2.data_ptr(), M, N, s), "mi_bn_mish_bwd")
                grads[4 * i + 2], grads[4 * i + 3] = dgb[0], dgb[1]
                if need_b:                # dbias = sum_m dz: joined like an un-normalised layer's (al = 1)
                    scratch = torch.empty((3, N), dtype=torch.float32, device=dev)
                    _lib.check(lib.mi_tail_bn_finalize_bwd_a(part2.data_ptr(), MT, M, N, None, ones.data_ptr(), None, None,
                                                             scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(),
                                                             None, 0, None, None, 1, dbias.data_ptr(), s), "mi_tail_bn_finalize_bwd_a")
            else:
                dy = torch.empty((M, N), dtype=torch.float32, device=dev)
                if use_bn == 2:
                    _lib.check(lib.mi_col_act_bwd(g.data_ptr(), Z.data_ptr(), N, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), 1,
                                                  dy.data_ptr(), part.data_ptr(), M, N, s), "mi_col_act_bwd")
                    finalize(part, not training, need_b and not training, True)
                    grads[4 * i + 2], grads[4 * i + 3] = dgb[0], dgb[1]
                else:
                    _lib.check(lib.mi_col_act_bwd(g.data_ptr(), Z.data_ptr(), N, None, None, _lib.ptr(b), 1, dy.data_ptr(),
                                                  part.data_ptr(), M, N, s), "mi_col_act_bwd")
                    finalize(part, True, need_b, False)
                    dz_plain = dy         # no normalisation: dz = dy
            if need_b:
                # under a training-mode BatchNorm BEHIND the Linear (use_bn 2) the batch mean is removed: exactly zero
                grads[4 * i + 1] = dbias if dbias is not None else torch.zeros((N,), dtype=torch.float32, device=dev)
            dz_keep = None
            if need_dx:
                dx = torch.empty((M, K), dtype=torch.float32, device=dev)
                if dz_plain is not None:
                    _lib.check(lib.mi_tail_dgrad_gemm_s(dz_plain.data_ptr(), None, N, None, None, None, None, W.data_ptr(), K, None, K,
                                                        None, None, None, 0.0, None, dx.data_ptr(), K, None, 0, None, M, N, K, None, s),
                               "mi_tail_dgrad_gemm_s")
                else:
                    dz_keep = torch.empty((M, N), dtype=torch.float32, device=dev) if need_W else None
                    _lib.check(lib.mi_tail_dgrad_gemm_s(dy.data_ptr(), Z.data_ptr(), N, c[0].data_ptr(), dzc[0].data_ptr(),
                                                        dzc[1].data_ptr(), dzc[2].data_ptr(), W.data_ptr(), K, None, K, None, None,
                                                        None, 0.0, None, dx.data_ptr(), K, None, 0, _lib.ptr(dz_keep), M, N, K, None, s),
                               "mi_tail_dgrad_gemm_s")
            if need_W:
                dz = dz_plain if dz_plain is not None else dz_keep
                if dz is None:            # no input-gradient product computed it on the fly: materialise
                    dz = torch.empty((M, N), dtype=torch.float32, device=dev)
                    _lib.check(lib.mi_bn_dz(dy.data_ptr(), Z.data_ptr(), N, c[0].data_ptr(), dzc[0].data_ptr(), dzc[1].data_ptr(),
                                            dzc[2].data_ptr(), dz.data_ptr(), M, N, s), "mi_bn_dz")
                dW = torch.zeros((N, K), dtype=torch.float32, device=dev)          # split-K slices meet in atomics
                later.append(dict(A=dz, B=a_in, C=dW, M=N, N=K, K=M, lda=N, ldb=K, ldc=K))
                grads[4 * i] = dW
            if need_dx:
                g = dx
        _kernels.gemm_multi(later, transA=True)
        return (None, None, None, g if need[3] else None, *grads)


def run_mish_mlp(plan: List[_L], use_bn: int, x: torch.Tensor) -> torch.Tensor:
    params = []
    for L in plan:
        params += [L.lin.weight, L.lin.bias, L.bn.weight if L.bn is not None else None, L.bn.bias if L.bn is not None else None]
    return MishMlpFn.apply(plan, use_bn, torch.is_grad_enabled(), x, *params)
