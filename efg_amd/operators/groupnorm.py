"""GroupNorm over channels-last maps, libefg_hip.so (csrc/batchnorm.hip, `efg_gn_*`).

`group_norm_nhwc(x, gn)` == `gn(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)` for an `nn.GroupNorm` and a contiguous
[B, H, W, C] (or [B, L, C]) tensor: the input projection in front of the transformer ($CQ/voxel_detr.py:43-51) is a
1x1 convolution -- a GEMM on the channels-last BEV map -- followed by GroupNorm(32, 256), and the encoder reads the
result as [B, H*W, C] tokens.  ATen's GroupNorm is NCHW-only on the GPU: around it the step paid four 72 MB
transposing copies (two forward, two backward)."""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L


class GroupNormNHWCFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        b, c = x.shape[0], x.shape[-1]
        x3 = x.contiguous().view(b, -1, c)
        rows = x3.shape[1]
        y = torch.empty_like(x3)
        mean = torch.empty(b * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws_bytes = L.lib().efg_gn_workspace_bytes(b, c)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        L.check(L.lib().efg_gn_forward_f32(L.ptr(x3), L.ptr(weight.contiguous()), L.ptr(bias.contiguous()), float(eps),
                                           b, rows, c, groups, L.ptr(y), L.ptr(mean), L.ptr(rstd), L.ptr(ws), ws_bytes,
                                           L.stream()))
        ctx.save_for_backward(x3, weight, mean, rstd)
        ctx.groups = groups
        return y.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x3, weight, mean, rstd = ctx.saved_tensors
        b, rows, c = x3.shape
        dy3 = dy.contiguous().view(b, rows, c)
        dx = torch.empty_like(x3)
        dweight = torch.empty(c, dtype=torch.float32, device=x3.device)
        dbias = torch.empty_like(dweight)
        ws_bytes = L.lib().efg_gn_workspace_bytes(b, c)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x3.device)
        L.check(L.lib().efg_gn_backward_f32(L.ptr(dy3), L.ptr(x3), L.ptr(weight.contiguous()), L.ptr(mean), L.ptr(rstd),
                                            b, rows, c, ctx.groups, L.ptr(dx), L.ptr(dweight), L.ptr(dbias), L.ptr(ws),
                                            ws_bytes, L.stream()))
        return dx.view(dy.shape), dweight, dbias, None, None


def fusable(x, gn):
    c = x.shape[-1]
    return (os.environ.get("EFG_FUSED_GN", "1") != "0" and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3
            and isinstance(gn, torch.nn.GroupNorm) and gn.affine and gn.num_channels == c and c % 4 == 0 and c <= 1024
            and (c // gn.num_groups) % 4 == 0 and x.shape[0] <= 65535 and x.numel() > 0)


def group_norm_nhwc(x, gn):
    """x [B, ..., C] channels-last -> same shape; statistics per (sample, group) over everything in between."""
    L.require_gpu(x)
    if not fusable(x, gn):
        raise RuntimeError("group_norm_nhwc: unsupported configuration (C %% 4, C / groups %% 4, fp32 required)")
    return GroupNormNHWCFunction.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps)
