"""Losses around the path.  `BCEWithLogitsLoss` is torch.nn.BCEWithLogitsLoss
(reduction="mean", the reference trainer's criterion, src/trainer/deepfm.py:32,51) as ONE HIP launch
each way instead of ~8 elementwise/reduction launches — at B=4096 every launch is microseconds.
`bpr_loss` restates src/losses.py:6-22 (tiny; stays in PyTorch ops)."""
import torch
from torch import nn

from . import _kernels, _lib


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        dev = _lib.require_gpu(logits, target)
        x = _kernels._f32c(logits).view(-1)
        y = _kernels._f32c(target).view(-1)
        if x.numel() != y.numel():
            raise ValueError("logits and target must have the same number of elements")
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().mi_bce_logits_fwd(x.data_ptr(), y.data_ptr(), loss.data_ptr(), x.numel(),
                                                 _lib.stream_ptr(dev)), "mi_bce_logits_fwd")
        ctx.save_for_backward(x, y)
        ctx.shape = tuple(logits.shape)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _kernels._f32c(g).view(1)
        dx = torch.empty_like(x)
        _lib.check(_lib.load().mi_bce_logits_bwd(x.data_ptr(), y.data_ptr(), g.data_ptr(), dx.data_ptr(), x.numel(),
                                                 _lib.stream_ptr(x.device)), "mi_bce_logits_bwd")
        return dx.view(ctx.shape), None


class BCEWithLogitsLoss(nn.Module):
    def forward(self, logits, target):
        return _BCEFn.apply(logits, target.float())


def bpr_loss(user_embs, pos_embs, neg_embs):
    y_hat_pos = (user_embs * pos_embs).sum(1)
    y_hat_neg = (user_embs * neg_embs).sum(1)
    return -torch.nn.functional.logsigmoid(y_hat_pos - y_hat_neg).mean()
