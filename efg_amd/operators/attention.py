"""Multi-head attention cores over the fused in-projection outputs (csrc/attention.hip).

`self_attention_qkv(qkv [B, S, 3*C], heads) -> [B, S, C]` and `cross_attention_kv(q [B, Sq, C], kv [B, Sk, 2*C], heads)
-> [B, Sq, C]` = softmax(Q K^T / sqrt(C / heads)) V per head: the part of nn.MultiheadAttention between its in- and
out-projection (TrajectoryFormer's point encoder, $TF/modules/transformer.py:44-92: `self_attn` over a hypothesis' 128
points, `point_attn` of its summary token against them).  On the GPU, for fp32, 64-wide heads and up to 128 tokens they
run the HIP kernels (one workgroup per (sequence, head), exact-fp32 MFMA, probabilities never leave the registers;
q / k / v are read in place from the projection outputs and the backward writes their gradients into tensors of the
same layout).  Other shapes and host tensors take `F.scaled_dot_product_attention`.  `EFG_ATTENTION=0` forces the
latter (A/B runs)."""
import math
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib as L

HEAD_DIM, MAX_SEQ = 64, 128
_SCALE = 1.0 / math.sqrt(HEAD_DIM)


def _forward(q, q_strides, k_off, v_off, kv, kv_strides, b, sq, sk, heads):
    out = torch.empty(b, sq, heads * HEAD_DIM, dtype=torch.float32, device=kv.device)
    lse = torch.empty(b, heads, sq, dtype=torch.float32, device=kv.device)
    L.check(L.lib().efg_attention_fwd_f32(q.data_ptr(), *q_strides, kv.data_ptr() + 4 * k_off, kv.data_ptr() + 4 * v_off,
                                          *kv_strides, b, sq, sk, heads, _SCALE, L.ptr(out), L.ptr(lse), L.stream()))
    return out, lse


def _backward(q, q_strides, k_off, v_off, kv, kv_strides, out, lse, grad, b, sq, sk, heads, dq, dkv):
    L.check(L.lib().efg_attention_bwd_f32(q.data_ptr(), *q_strides, kv.data_ptr() + 4 * k_off, kv.data_ptr() + 4 * v_off,
                                          *kv_strides, L.ptr(out), L.ptr(lse), L.ptr(grad), b, sq, sk, heads, _SCALE,
                                          dq.data_ptr(), dkv.data_ptr() + 4 * k_off, dkv.data_ptr() + 4 * v_off, L.stream()))


class _SelfAttention(Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = qkv.contiguous()
        b, s, c3 = qkv.shape
        c = c3 // 3
        out, lse = _forward(qkv, (s * c3, c3), c, 2 * c, qkv, (s * c3, c3), b, s, s, heads)
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, grad):
        qkv, out, lse = ctx.saved_tensors
        b, s, c3 = qkv.shape
        c = c3 // 3
        dqkv = torch.empty_like(qkv)
        _backward(qkv, (s * c3, c3), c, 2 * c, qkv, (s * c3, c3), out, lse, grad.contiguous(), b, s, s, ctx.heads, dqkv, dqkv)
        return dqkv, None


class _CrossAttention(Function):
    @staticmethod
    def forward(ctx, q, kv, heads):
        q, kv = q.contiguous(), kv.contiguous()
        b, sq, c = q.shape
        sk = kv.shape[1]
        out, lse = _forward(q, (sq * c, c), 0, c, kv, (sk * 2 * c, 2 * c), b, sq, sk, heads)
        ctx.save_for_backward(q, kv, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, grad):
        q, kv, out, lse = ctx.saved_tensors
        b, sq, c = q.shape
        sk = kv.shape[1]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        _backward(q, (sq * c, c), 0, c, kv, (sk * 2 * c, 2 * c), out, lse, grad.contiguous(), b, sq, sk, ctx.heads, dq, dkv)
        return dq, dkv, None


def _eligible(t, heads, parts):
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.shape[-1] == parts * heads * HEAD_DIM
            and 1 <= t.shape[1] <= MAX_SEQ and t.shape[0] > 0 and os.environ.get("EFG_ATTENTION", "1") != "0")


def fused(qkv, heads):
    return _eligible(qkv, heads, 3)


def fused_cross(q, kv, heads):
    return _eligible(q, heads, 1) and _eligible(kv, heads, 2) and q.shape[0] == kv.shape[0]


def _sdpa(q, k, v, heads):
    b, sq, c = q.shape

    def split(t):
        return t.reshape(b, t.shape[1], heads, c // heads).transpose(1, 2)

    return F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(b, sq, c)


def self_attention_qkv(qkv, heads):
    if fused(qkv, heads):
        return _SelfAttention.apply(qkv, heads)
    return _sdpa(*qkv.chunk(3, dim=-1), heads)


def cross_attention_kv(q, kv, heads):
    if fused_cross(q, kv, heads):
        return _CrossAttention.apply(q, kv, heads)
    return _sdpa(q, *kv.chunk(2, dim=-1), heads)


# ---- long sequences, 32-wide heads, boolean mask (the decoder's self-attention in ConQueR / Voxel-DETR) ------------------------
LONG_HEAD_DIM = 32


def pack_mask(attn_mask):
    """bool [S, S] (True = not attended) -> int32 [S, ceil(S / 32)] bit rows for the kernels; None -> None."""
    if attn_mask is None:
        return None
    s = attn_mask.shape[-1]
    words = (s + 31) // 32
    m = torch.zeros(attn_mask.shape[0], words * 32, dtype=torch.int64, device=attn_mask.device)
    m[:, :s] = attn_mask.to(torch.int64)
    shifts = torch.arange(32, device=attn_mask.device, dtype=torch.int64)
    packed = (m.view(-1, words, 32) << shifts).sum(-1)          # < 2^32
    return torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32).contiguous()


class _LongSelfAttention(Function):
    """qk [B, S, 2C] (q | k of one fused projection), v [B, S, C], mask bits -> [B, S, C]."""

    @staticmethod
    def _strides(qk, v):
        b, s, c2 = qk.shape
        c = c2 // 2
        return (qk.data_ptr(), s * c2, c2, qk.data_ptr() + 4 * c, s * c2, c2, v.data_ptr(), s * c, c)

    @staticmethod
    def forward(ctx, qk, v, mask_bits, heads):
        qk, v = qk.contiguous(), v.contiguous()
        b, s, c = v.shape
        out = torch.empty(b, s, c, dtype=torch.float32, device=v.device)
        lse = torch.empty(b, heads, s, dtype=torch.float32, device=v.device)
        scale = 1.0 / math.sqrt(LONG_HEAD_DIM)
        L.check(L.lib().efg_attention_long_fwd_f32(*_LongSelfAttention._strides(qk, v), L.ptr(mask_bits),
                                                   0 if mask_bits is None else mask_bits.shape[1], b, s, heads, scale,
                                                   L.ptr(out), L.ptr(lse), L.stream()))
        ctx.save_for_backward(qk, v, out, lse, mask_bits)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, grad):
        qk, v, out, lse, mask_bits = ctx.saved_tensors
        b, s, c = v.shape
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        scale = 1.0 / math.sqrt(LONG_HEAD_DIM)
        L.check(L.lib().efg_attention_long_bwd_f32(*_LongSelfAttention._strides(qk, v), L.ptr(mask_bits),
                                                   0 if mask_bits is None else mask_bits.shape[1], L.ptr(out), L.ptr(lse),
                                                   L.ptr(grad.contiguous()), b, s, ctx.heads, scale, dqk.data_ptr(),
                                                   dqk.data_ptr() + 4 * c, dv.data_ptr(), L.ptr(torch.empty_like(lse)),
                                                   L.stream()))
        return dqk, dv, None, None


def fused_long(x, heads):
    """x: the [B, S, C] input of the attention (batch first)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] == heads * LONG_HEAD_DIM
            and x.shape[0] * heads < 65536 and x.shape[0] > 0 and x.shape[1] > 0
            and os.environ.get("EFG_ATTENTION", "1") != "0")


def long_self_attention(qk, v, mask_bits, heads):
    return _LongSelfAttention.apply(qk, v, mask_bits, heads)
