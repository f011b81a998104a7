"""Weighted 2-D cross entropy on the GPU kernels.

`CrossEntropyLoss2d(weight)` has the interface of /root/reference/train/main.py:72-80 and
computes what `torch.nn.CrossEntropyLoss(weight)` computes at train/main.py:121,202:

    L = sum_p w[t_p] * (-log softmax(z_p)[t_p]) / sum_p w[t_p]

Numerator and denominator are produced separately (erf_weighted_ce_fwd).  With `group` set, both
are all-reduced over the process group before the division, which reproduces nn.DataParallel's
loss over the gathered global batch (per-rank means would weight the shards unequally); the
backward pass then scales every rank's dlogits by 1/global-denominator, so summing the ranks'
parameter gradients gives the global-batch gradient.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import call, query
from .dp import global_weighted_mean


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _WeightedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, group):
        ops.require_gpu(logits, "CrossEntropyLoss2d")
        ctx.flat = logits.dim() == 2                 # [N, K] classifier logits (imagenet/main.py:81): one "pixel" per image
        if ctx.flat:
            logits, target = logits[:, :, None, None], target[:, None, None]
        logits = logits.contiguous()
        target = target.contiguous()
        if target.dtype != torch.int64:
            raise TypeError("targets must be int64 class indices")
        n, c, h, w = logits.shape
        if target.shape != (n, h, w):
            raise ValueError(f"expected target of shape {(n, h, w)}, got {tuple(target.shape)}")
        dev = logits.device
        wt = None if weight is None else weight.detach().to(dev, torch.float32).contiguous()
        out3 = ops.T.weighted_ce_fwd(logits, target, wt)
        num, den = global_weighted_mean(out3[0], out3[1], group)
        ctx.save_for_backward(logits, target, den.reshape(1))
        ctx.weight = wt
        return num / den

    @staticmethod
    def backward(ctx, gout):
        logits, target, den = ctx.saved_tensors
        n, c, h, w = logits.shape
        coef = (gout.reshape(1).float() / den).contiguous()
        dlogits = ops.T.weighted_ce_bwd(logits, target, ctx.weight, coef)
        return (dlogits[:, :, 0, 0] if ctx.flat else dlogits), None, None, None


def weighted_cross_entropy(logits, target, weight=None, group=None):
    return _WeightedCE.apply(logits, target, weight, group)


class CrossEntropyLoss2d(torch.nn.Module):

    def __init__(self, weight=None, group=None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32))
        self.group = group

    def forward(self, outputs, targets):
        return weighted_cross_entropy(outputs, targets, self.weight, self.group)
