"""Fused BatchNorm1d (+ residual) (+ ReLU) over sparse-tensor features (csrc/batchnorm.hip).

`bn_act(x, bn, relu=..., residual=...)` == `relu(bn(x) + residual)` for an `nn.BatchNorm1d` in training mode --
the norm / activation steps of the sparse backbone (efg/modeling/backbones/sparse_net.py:85-95,120-165), two
kernels forward and two backward instead of PyTorch's batch_norm + add + ReLU chain.  The module (and its
state-dict entries, running statistics, num_batches_tracked) is the ordinary nn.BatchNorm1d."""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L

class BatchNormActFunction(Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu):
        x = x.contiguous()
        m, c = x.shape
        res = residual.contiguous() if residual is not None else None
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        ws_bytes = L.lib().efg_bn_workspace_bytes(c)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        L.check(L.lib().efg_bn_forward_f32(L.ptr(x), L.ptr(res), L.ptr(weight), L.ptr(bias), L.ptr(running_mean),
                                           L.ptr(running_var), L.ptr(num_batches_tracked), float(momentum), float(eps),
                                           m, c, 1 if relu else 0, L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(ws),
                                           ws_bytes, L.stream()))
        ctx.save_for_backward(x, y, weight, mean, invstd)
        ctx.relu, ctx.has_residual = bool(relu), residual is not None
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, num_batches_tracked) if t is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, weight, mean, invstd = ctx.saved_tensors
        m, c = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_residual else None
        dweight = torch.empty(c, dtype=torch.float32, device=x.device)
        dbias = torch.empty(c, dtype=torch.float32, device=x.device)
        ws_bytes = L.lib().efg_bn_workspace_bytes(c)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        L.check(L.lib().efg_bn_backward_f32(L.ptr(dy), L.ptr(x), L.ptr(y), L.ptr(weight), L.ptr(mean), L.ptr(invstd), m,
                                            c, 1 if ctx.relu else 0, L.ptr(dx), L.ptr(dres), L.ptr(dweight),
                                            L.ptr(dbias), L.ptr(ws), ws_bytes, L.stream()))
        return dx, dres, dweight, dbias, None, None, None, None, None, None


def fusable(bn, x):
    return (os.environ.get("EFG_FUSED_BN", "1") != "0" and isinstance(bn, torch.nn.BatchNorm1d) and bn.training
            and bn.affine and bn.track_running_stats and bn.momentum is not None and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 2 and x.shape[1] % 4 == 0
            and x.shape[1] <= 1024)


def bn_act(x, bn, relu=False, residual=None):
    """relu?(bn(x) + residual?) for features [M, C]."""
    if not fusable(bn, x):  # eval mode, host tensors (CPU tests), exotic shapes: the PyTorch modules
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    return BatchNormActFunction.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      bn.num_batches_tracked, bn.momentum, bn.eps, relu)


def fusable_nhwc(bn, x):
    """A training-mode BatchNorm2d over a channels-last map: per-channel statistics over N*H*W rows, i.e. the [M, C] row
    problem the kernels above solve (the dense BEV stacks of CenterPoint: RPN, shared conv, task heads)."""
    return (os.environ.get("EFG_FUSED_BN", "1") != "0" and type(bn) is torch.nn.BatchNorm2d and bn.training and bn.affine
            and bn.track_running_stats and bn.momentum is not None and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4 and x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 1024
            and x.shape[0] * x.shape[2] * x.shape[3] >= 2 and x.is_contiguous(memory_format=torch.channels_last))


def bn_act_nhwc(x, bn, relu=False):
    """relu?(bn(x)) for a channels-last [B, C, H, W] map, result channels-last: one fused pass each way instead of
    MIOpen's batch norm + a separate ReLU (and its threshold_backward)."""
    b, c, h, w = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(b * h * w, c)   # a view: channels-last memory is [B*H*W, C] row-major
    y = BatchNormActFunction.apply(rows, None, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, bn.momentum, bn.eps, relu)
    return y.view(b, h, w, c).permute(0, 3, 1, 2)


def run_sequential(seq, x):
    """`seq(x)` for an nn.Sequential, with every (BatchNorm2d[, ReLU]) pair that qualifies evaluated by `bn_act_nhwc`.
    The modules (and so the parameter names and the eval-mode path) are the reference's."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if fusable_nhwc(m, x):
            relu = i + 1 < len(mods) and type(mods[i + 1]) is torch.nn.ReLU
            x = bn_act_nhwc(x, m, relu)
            i += 2 if relu else 1
        else:
            x = m(x)
            i += 1
    return x
