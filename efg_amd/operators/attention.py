"""Multi-head self-attention core over the fused in-projection output.

`self_attention_qkv(qkv [B, S, 3*C], heads) -> [B, S, C]` = softmax(Q K^T / sqrt(C / heads)) V per head, the part of
nn.MultiheadAttention between its in- and out-projection (TrajectoryFormer's point encoder,
$TF/modules/transformer.py:44-92).  On the GPU, for fp32, 64-wide heads and up to 128 tokens it runs csrc/attention.hip
(one workgroup per (sequence, head), exact-fp32 MFMA, probabilities never leave the registers; backward writes the
gradient of `qkv` as ONE tensor).  Other shapes and host tensors take `F.scaled_dot_product_attention`.
`EFG_ATTENTION=0` forces the latter (A/B runs)."""
import math
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib as L

HEAD_DIM, MAX_SEQ = 64, 128


class _SelfAttention(Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = qkv.contiguous()
        b, s, _ = qkv.shape
        out = torch.empty(b, s, heads * HEAD_DIM, dtype=torch.float32, device=qkv.device)
        lse = torch.empty(b, heads, s, dtype=torch.float32, device=qkv.device)
        L.check(L.lib().efg_attention_fwd_f32(L.ptr(qkv), b, s, heads, 1.0 / math.sqrt(HEAD_DIM), L.ptr(out), L.ptr(lse),
                                              L.stream()))
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, grad):
        qkv, out, lse = ctx.saved_tensors
        b, s, _ = qkv.shape
        grad = grad.contiguous()
        dqkv = torch.empty_like(qkv)
        L.check(L.lib().efg_attention_bwd_f32(L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(grad), b, s, ctx.heads,
                                              1.0 / math.sqrt(HEAD_DIM), L.ptr(dqkv), L.stream()))
        return dqkv, None


def fused(qkv, heads):
    return (qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 3 and qkv.shape[-1] == 3 * heads * HEAD_DIM
            and 1 <= qkv.shape[1] <= MAX_SEQ and qkv.shape[0] > 0 and os.environ.get("EFG_ATTENTION", "1") != "0")


def self_attention_qkv(qkv, heads):
    if fused(qkv, heads):
        return _SelfAttention.apply(qkv, heads)
    b, s, c3 = qkv.shape
    q, k, v = (t.reshape(b, s, heads, c3 // 3 // heads).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, s, c3 // 3)
