"""3 x 3 dense convolution as fixed-order GEMMs over the zero-padded channels-last map (EFG_DETERMINISTIC=1).

The one dense 3 x 3 convolution of the path (FPN `fpn_output3`, 256 -> 256 at 188 x 188, efg/modeling/backbones/fpn.py:47)
normally runs on MIOpen, whose solver is picked by TIMING (`cudnn.benchmark`, engine.py): its fp32 weight-gradient solver
splits the reduction with float atomics on every box, and on a box whose host is loaded while the find runs the forward /
data-gradient pick can be an atomic one too -- then two identical steps differ from this layer's OUTPUT on
(`scripts/ubench/determinism_probe.py --trace`; MIOpen's own deterministic mode picks a naive solver: 10x the whole step).
With `EFG_DETERMINISTIC=1` the convolution does not touch MIOpen.  A shifted window of a zero-padded row-major map is a
CONTIGUOUS row range, so with off(ky, kx) = (ky - 1) (W + 2) + (kx - 1) over the rows r of the padded maps

    Y[r]             = b + sum_k  X_pad[r + off_k]  W_k^T          nine `addmm_` into one buffer, in tap order
    dX[r]            =     sum_k  G_pad[r - off_k]  W_k            the same, on the padded gradient
    dW[:, :, ky, kx] =     sum_r  G_pad[r]^T  X_pad[r + off_k]     nine `operators.linear.weight_grad` (16 fixed row chunks)

(the border rows of Y and dX are computed and dropped; the padding rows of G are zero, so every out-of-image pair drops out
of dW).  hipBLASLt does not split K = 256, so each product has one summation order.  +0.5 ms per step, hence opt-in."""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .linear import weight_grad


def deterministic_mode():
    return os.environ.get("EFG_DETERMINISTIC", "0") == "1"


def _pad_rows(t, front):
    """[B, C, H, W] -> zero-padded channels-last rows [front + rows + front, C]; returns (buffer, r, rows)."""
    b, c, h, w = t.shape
    r = b * (h + 2) * (w + 2)
    rows = (r + 15) // 16 * 16
    buf = torch.zeros((front + rows + front, c), dtype=t.dtype, device=t.device)
    buf[front:front + r].view(b, h + 2, w + 2, c)[:, 1:-1, 1:-1].copy_(t.permute(0, 2, 3, 1))
    return buf, r, rows


def _unpad_rows(flat, b, h, w):
    """rows of the padded map -> [B, C, H, W] (a channels-last view of the interior)."""
    return flat[:b * (h + 2) * (w + 2)].view(b, h + 2, w + 2, -1)[:, 1:-1, 1:-1].permute(0, 3, 1, 2)


_TAPS = [(ky, kx) for ky in range(3) for kx in range(3)]


def _taps_gemm(buf, rows, front, wp, mats, sign, bias=None):
    """sum_k buf[front + sign * off_k :][:rows] @ mats[k], accumulated in tap order into one [rows, N] buffer."""
    out = (bias.expand(rows, -1).contiguous() if bias is not None else
           torch.zeros((rows, mats[0].shape[1]), dtype=buf.dtype, device=buf.device))
    for (ky, kx), m in zip(_TAPS, mats):
        s = front + sign * ((ky - 1) * wp + (kx - 1))
        out.addmm_(buf[s:s + rows], m)
    return out


def wgrad_3x3(x, gy, xb=None):
    """x [B, Ci, H, W], gy [B, Co, H, W] (stride 1, padding 1) -> dW [Co, Ci, 3, 3], fixed summation order."""
    b, ci, h, w = gy.shape[0], (x.shape[1] if xb is None else xb.shape[1]), gy.shape[2], gy.shape[3]
    co = gy.shape[1]
    wp = w + 2
    front = wp + 1
    if xb is None:
        xb, _, _ = _pad_rows(x, front)
    gb, _, rows = _pad_rows(gy, 0)
    parts = []
    for ky, kx in _TAPS:
        s = front + (ky - 1) * wp + (kx - 1)
        parts.append(weight_grad(xb[s:s + rows], gb))     # [Co, Ci]
    return torch.stack(parts, dim=-1).view(co, ci, 3, 3)


class Conv3x3Function(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        b, _, h, w = x.shape
        front = w + 3
        xb, _, rows = _pad_rows(x, front)
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.geom = (b, h, w)
        mats = [weight[:, :, ky, kx].t() for ky, kx in _TAPS]              # [Ci, Co] each
        return _unpad_rows(_taps_gemm(xb, rows, front, w + 2, mats, +1, bias), b, h, w)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xb, weight = ctx.saved_tensors
        b, h, w = ctx.geom
        front = w + 3
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gbuf, _, rows = _pad_rows(gy, front)
            mats = [weight[:, :, ky, kx] for ky, kx in _TAPS]              # [Co, Ci] each
            gx = _unpad_rows(_taps_gemm(gbuf, rows, front, w + 2, mats, -1), b, h, w)
        if ctx.needs_input_grad[1]:
            gw = wgrad_3x3(None, gy, xb=xb).contiguous(memory_format=torch.channels_last if weight.is_contiguous(
                memory_format=torch.channels_last) and not weight.is_contiguous() else torch.contiguous_format)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb


def conv3x3(x, weight, bias):
    return Conv3x3Function.apply(x, weight, bias)


def _tap_rows(buf, start_row, rows, c):
    """[rows, 3c] view of the padded map: row r = the 3c consecutive floats starting at row start_row + r, i.e. the three
    kx taps of one ky side by side.  The rows OVERLAP (row stride c): the convolution's im2col block without a copy."""
    return buf.as_strided((rows, 3 * c), (c, 1), buf.storage_offset() + start_row * c)


class Conv3x3ArmFunction(Function):
    """The same convolution on the split-precision A/B arm (EFG_GEMM_ARM=bf16x3; never the default): per ky ONE product of
    the overlapping-row view above with the [3 Ci, Co] block of the weights (csrc/gemm_bf16x3.hip takes the row stride as an
    argument), three products per pass instead of MIOpen's implicit GEMM -- forward, data gradient and weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import gemm_bf16x3 as G

        b, ci, h, w = x.shape
        co = weight.shape[0]
        wp, front = w + 2, w + 3
        xb, _, rows = _pad_rows(x, front)
        wk = weight.detach().permute(2, 3, 1, 0).contiguous()                 # [ky][kx][ci][co]
        out = None
        for ky in range(3):
            y = G.gemm(_tap_rows(xb, front + (ky - 1) * wp - 1, rows, ci), G.pack(wk[ky], 3 * ci, co, co, 1), co,
                       bias=bias if ky == 0 else None)
            out = y if out is None else out.add_(y)
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.geom = (b, h, w)
        return _unpad_rows(out, b, h, w)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        from . import gemm_bf16x3 as G

        xb, weight = ctx.saved_tensors
        b, h, w = ctx.geom
        co, ci = weight.shape[:2]
        wp, front = w + 2, w + 3
        gbuf, _, rows = _pad_rows(gy, front)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # gx[p] = sum_k G[p - off_k] W_k: per ky the rows p - (ky - 1) wp + 1, .. , - 1 side by side, i.e. kx = 2, 1, 0
            wd = weight.detach().flip(3).permute(2, 3, 0, 1).contiguous()     # [ky][2 - kx][co][ci]
            acc = None
            for ky in range(3):
                y = G.gemm(_tap_rows(gbuf, front - (ky - 1) * wp - 1, rows, co), G.pack(wd[ky], 3 * co, ci, ci, 1), ci)
                acc = y if acc is None else acc.add_(y)
            gx = _unpad_rows(acc, b, h, w)
        if ctx.needs_input_grad[1]:
            g_rows = gbuf[front:front + rows]
            gw = torch.empty((co, ci, 3, 3), dtype=weight.dtype, device=weight.device)
            for ky in range(3):
                d = G.wgrad(g_rows, _tap_rows(xb, front + (ky - 1) * wp - 1, rows, ci))   # [co, 3 ci] = (kx, ci)
                gw[:, :, ky, :] = d.view(co, 3, ci).permute(0, 2, 1)
            if weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous():
                gw = gw.contiguous(memory_format=torch.channels_last)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb


def conv3x3_arm(x, weight, bias):
    return Conv3x3ArmFunction.apply(x, weight, bias)


def arm_covers(x, weight):
    """The arm's products want 16-byte rows and whole 64-channel blocks on both sides."""
    return weight.shape[0] % 64 == 0 and weight.shape[1] % 64 == 0 and x.dtype == torch.float32
