"""`Linear`: nn.Linear whose backward takes the bias gradient from libefg_hip.so (csrc/colsum.hip).

The matrix products stay where they were -- `addmm` forward, `grad @ W` and `xᵀ @ grad` backward, the very calls
autograd makes for `F.linear`, so the tuned hipBLASLt solutions (efg_amd/tuned/gemm_gfx950.csv) keep applying --
only `grad_output.sum(0)` is replaced: ATen runs the ~100 such reductions of a training step at 1.8 TB/s on the
70 688-token encoder sequence and at 15 us apiece on the decoder's few thousand rows (1.7 ms per step together).
Same parameters and state-dict names as nn.Linear ($CQ/transformer.py:215-243,273-317, $CQ/modules/blocks.py:5-17,
$CQ/modules/box_attention.py:31-40)."""
import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L


def column_sum(x2):
    """x2 [rows, cols] fp32 on the GPU (rows may be strided) -> [cols], deterministic."""
    L.require_gpu(x2)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows, cols = x2.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2.device)
    ws_bytes = L.lib().efg_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x2.device)
    L.check(L.lib().efg_colsum_f32(x2.data_ptr(), rows, cols, x2.stride(0) if rows > 1 else cols, L.ptr(out), L.ptr(ws),
                                   ws_bytes, L.stream()))
    return out


class LinearFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        return torch.addmm(bias, x2, weight.t()).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        x2, weight = ctx.saved_tensors
        g2 = grad.reshape(-1, grad.shape[-1])
        gx = g2.mm(weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        gw = x2.t().mm(g2).t() if ctx.needs_input_grad[1] else None
        gb = column_sum(g2) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def linear(x, weight, bias=None):
    """F.linear; on the GPU, in training, with the bias gradient from the HIP column sum."""
    if (x.is_cuda and bias is not None and x.dtype == torch.float32 and torch.is_grad_enabled()
            and bias.requires_grad and x.numel() > 0 and os.environ.get("EFG_FUSED_LINEAR", "1") != "0"):
        return LinearFunction.apply(x, weight, bias)
    return F.linear(x, weight, bias)  # host tensors (the CPU tests), inference: plain PyTorch, same math


class Linear(nn.Linear):
    def forward(self, x):
        return linear(x, self.weight, self.bias)
