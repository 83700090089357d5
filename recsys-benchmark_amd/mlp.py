"""The MLP tail of DeepFM / DCN: (Linear, [BatchNorm1d], ReLU, Dropout) x k +
Linear — reference: src/models/deepfm.py:53-66,100-102 and src/models/dcn.py:56-66.

The modules stay ordinary nn.Linear / nn.BatchNorm1d / nn.ReLU / nn.Dropout inside the same
nn.Sequential (state_dict keys unchanged); `run_tail` walks the Sequential and executes every
(Linear, [BatchNorm1d], ReLU, [Dropout]) group as: the contraction on hipBLASLt/rocBLAS through
PyTorch (a real GEMM), then ONE fused BN+ReLU+Dropout HIP pass each way (mi_bn_relu_dropout_*),
with the bias gradient from the library's column-sum kernel.  Anything that does not match the
pattern is run as the plain module.
"""
from typing import Dict, List

import torch
from torch import nn

from . import _kernels, _lib

_seeds: Dict[int, torch.Tensor] = {}


def _seed_word(dev: torch.device) -> torch.Tensor:
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    w = _seeds.get(i)
    if w is None:
        w = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFF], dtype=torch.int64, device=torch.device("cuda", i))
        _seeds[i] = w
    return w


class _LinearFn(torch.autograd.Function):
    """z = x W^T + b on rocBLAS/hipBLASLt; the bias gradient by the library's column sum."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return torch.addmm(b, x, W.t()) if b is not None else x @ W.t()

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous()
        dx = g @ W if ctx.needs_input_grad[0] else None
        dW = g.t() @ x if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            M, N = g.shape
            db = torch.zeros((N,), dtype=torch.float32, device=g.device)
            _lib.check(_lib.load().mi_colsum(g.data_ptr(), N, None, 0, db.data_ptr(), M, N, _lib.stream_ptr(g.device)),
                       "mi_colsum")
        return dx, dW, db


class _BNReLUDropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, has_bn, training, momentum, eps, p, seed, salt):
        dev = _lib.require_gpu(z)
        z = _kernels._f32c(z)
        M, N = z.shape
        bn_train = bool(has_bn and training)
        drop = bool(training and p > 0.0)
        y = torch.empty_like(z)
        keep = torch.empty((M, N), dtype=torch.uint8, device=dev) if drop else None
        stats = torch.zeros((2, N), dtype=torch.float32, device=dev) if bn_train else None
        save = torch.empty((2, N), dtype=torch.float32, device=dev) if has_bn else None
        _lib.check(
            _lib.load().mi_bn_relu_dropout_fwd(
                z.data_ptr(), N, M, N, int(has_bn), int(training), _lib.ptr(gamma), _lib.ptr(beta),
                _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum), float(eps), float(p if drop else 0.0),
                _lib.ptr(seed), int(salt), _lib.ptr(stats), y.data_ptr(), _lib.ptr(keep),
                save[0].data_ptr() if has_bn else None, save[1].data_ptr() if has_bn else None, _lib.stream_ptr(dev)),
            "mi_bn_relu_dropout_fwd",
        )
        ctx.save_for_backward(z, gamma, beta, keep, save)
        ctx.meta = (M, N, bool(has_bn), bool(training), float(p if drop else 0.0))
        return y

    @staticmethod
    def backward(ctx, dy):
        z, gamma, beta, keep, save = ctx.saved_tensors
        M, N, has_bn, training, p = ctx.meta
        dev = z.device
        dy = _kernels._f32c(dy)
        dz = torch.empty_like(z)
        dgb = torch.zeros((2, N), dtype=torch.float32, device=dev) if has_bn else None
        _lib.check(
            _lib.load().mi_bn_relu_dropout_bwd(
                dy.data_ptr(), z.data_ptr(), N, M, N, int(has_bn), int(training), _lib.ptr(keep), p, _lib.ptr(gamma),
                _lib.ptr(beta), save[0].data_ptr() if has_bn else None, save[1].data_ptr() if has_bn else None,
                _lib.ptr(dgb), dz.data_ptr(), _lib.stream_ptr(dev)),
            "mi_bn_relu_dropout_bwd",
        )
        dgamma = dgb[0] if (has_bn and gamma is not None and ctx.needs_input_grad[1]) else None
        dbeta = dgb[1] if (has_bn and beta is not None and ctx.needs_input_grad[2]) else None
        return dz, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def _groups(seq: nn.Sequential) -> List[List[nn.Module]]:
    """Split the Sequential into fusable (Linear, [BN], ReLU, [Dropout]) groups and single modules."""
    mods = list(seq)
    out, i = [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            j = i + 1
            bn = mods[j] if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d) else None
            if bn is not None:
                j += 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                j += 1
                dp = mods[j] if j < len(mods) and isinstance(mods[j], nn.Dropout) else None
                if dp is not None:
                    j += 1
                if bn is None or (bn.momentum is not None and bn.track_running_stats):
                    out.append(["fused", m, bn, dp])
                    i = j
                    continue
        out.append(["plain", m])
        i += 1
    return out


def run_tail(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    dev = x.device
    seed = _seed_word(dev)
    bumped = False
    for k, grp in enumerate(_groups(seq)):
        if grp[0] == "plain":
            m = grp[1]
            x = _LinearFn.apply(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
            continue
        _, lin, bn, dp = grp
        z = _LinearFn.apply(x, lin.weight, lin.bias)
        training_bn = bn is not None and bn.training
        p = dp.p if (dp is not None and dp.training) else 0.0
        if p > 0.0 and not bumped:
            seed.add_(1)          # one new dropout stream per tail pass (graph-capture safe)
            bumped = True
        if training_bn:
            bn.num_batches_tracked.add_(1)
        x = _BNReLUDropFn.apply(
            z, bn.weight if bn is not None else None, bn.bias if bn is not None else None,
            bn.running_mean if bn is not None else None, bn.running_var if bn is not None else None,
            bn is not None, bool(training_bn or p > 0.0) if bn is None else bool(bn.training),
            bn.momentum if bn is not None else 0.0, bn.eps if bn is not None else 0.0, p, seed, 7919 * (k + 1))
    return x
