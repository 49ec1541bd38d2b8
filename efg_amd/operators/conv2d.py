"""3 x 3 dense convolution whose WEIGHT GRADIENT is nine fixed-order GEMMs instead of MIOpen's `wrw` kernel.

MIOpen's fp32 weight-gradient solver for the one dense 3 x 3 convolution of the path (FPN `fpn_output3`, 256 -> 256 at
188 x 188, efg/modeling/backbones/fpn.py:47) splits its reduction with float atomics: that tensor's gradient is the
LAST thing that differs between two identical training steps (`scripts/ubench/determinism_probe.py`; MIOpen's own
deterministic mode picks a naive solver: 10x the whole step).  With `EFG_DETERMINISTIC=1` the convolution keeps MIOpen
for the forward and the data gradient and computes

    dW[:, :, ky, kx] = sum_r  G_pad[r]^T  X_pad[r + (ky - 1) (W + 2) + (kx - 1)]

over the rows of the zero-padded channels-last maps -- a shifted window of a padded row-major map is a CONTIGUOUS row
range, and the padding rows of G are zero, so every out-of-image pair drops out -- as nine products of
`operators.linear.weight_grad` (16 fixed row chunks, summed in order).  +0.2 ms per step, hence opt-in."""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .linear import weight_grad


def deterministic_mode():
    return os.environ.get("EFG_DETERMINISTIC", "0") == "1"


def wgrad_3x3(x, gy):
    """x [B, Ci, H, W], gy [B, Co, H, W] (stride 1, padding 1) -> dW [Co, Ci, 3, 3], fixed summation order."""
    b, ci, h, w = x.shape
    co = gy.shape[1]
    wp = w + 2
    r = b * (h + 2) * wp
    rows = (r + 15) // 16 * 16
    front = wp + 1
    xb = torch.zeros((front + rows + front, ci), dtype=x.dtype, device=x.device)
    gb = torch.zeros((rows, co), dtype=gy.dtype, device=gy.device)
    xb[front:front + r].view(b, h + 2, wp, ci)[:, 1:-1, 1:-1].copy_(x.permute(0, 2, 3, 1))
    gb[:r].view(b, h + 2, wp, co)[:, 1:-1, 1:-1].copy_(gy.permute(0, 2, 3, 1))
    parts = []
    for ky in range(3):
        for kx in range(3):
            s = front + (ky - 1) * wp + (kx - 1)
            parts.append(weight_grad(xb[s:s + rows], gb))     # [Co, Ci]
    return torch.stack(parts, dim=-1).view(co, ci, 3, 3)


class Conv3x3Function(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.conv2d(x, weight, bias, stride=1, padding=1)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            gw = wgrad_3x3(x, gy).contiguous(memory_format=torch.channels_last if weight.is_contiguous(
                memory_format=torch.channels_last) and not weight.is_contiguous() else torch.contiguous_format)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb


def conv3x3(x, weight, bias):
    return Conv3x3Function.apply(x, weight, bias)
