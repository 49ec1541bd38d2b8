"""The two halves of a Voxel-DETR encoder layer ($CQ/transformer.py:206-243) as ONE autograd node each.

    attention half:  norm1(src + out_proj(box_attention(value_proj(src), [logits | offsets](src + pos))))
    FFN half:        norm2(src + linear2(relu(linear1(src))))

Built from the same pieces as the module-by-module form -- the library GEMMs, csrc/box_fused.hip, csrc/layernorm.hip,
csrc/colsum.hip -- so the forward values are the same bits.  What the single node buys is in the backward: the layer
input `src` feeds three consumers in the attention half (residual, value projection, query projection) and two in the
FFN half, and autograd adds their gradients with separate element-wise kernels -- two and one additions of the
[B, 35 344, 256] gradient (72 MB each) per layer.  Here the LayerNorm backward's `dz` IS the residual gradient, and
each projection's data gradient is accumulated into it by its own GEMM (`addmm_`, beta = 1): no addition kernel, 144 MB
of traffic less per accumulation, and one Python autograd node per half instead of six / four.

The fused halves read the submodules' PARAMETERS and never call the submodules: forward / backward hooks registered on
`value_proj`, `out_proj`, `linear1`, `linear2`, `norm1`, `norm2` do not fire while the single-node form is in use (the same
holds for spconv.conv_bn_act and spconv.run_modules' conv + BatchNorm fusion).  EFG_FUSED_ENCODER=0 (EFG_FUSED_CONV_BN=0)
restores the module-by-module form for code that depends on such hooks.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._fuse import Ctx as _Ctx, pack as _pack, unpack as _unpack
from ..operators import box_attention_func as _baf
from ..operators import linear as _lin
from ..operators.layernorm import AddLayerNormFunction

_ENABLED = os.environ.get("EFG_FUSED_ENCODER", "1") != "0"


class EncoderAttentionHalf(Function):
    @staticmethod
    def forward(ctx, src, pos, ref_windows, v_shape, v_start, kidx, wv, bv, wa, ba, wb, bb, wo, bo, gamma, beta, eps,
                num_head, num_var):
        b, n, c = src.shape
        src2 = src.reshape(-1, c)
        q2 = (src + pos).reshape(-1, c) if pos is not None else src2
        value = torch.addmm(bv, src2, wv.t())
        wlo, blo = torch.cat((wa, wb), 0), torch.cat((ba, bb), 0)
        lo = torch.addmm(blo, q2, wlo.t())
        box = _Ctx()
        sampled = _baf.BoxAttnFusedFunction.forward(box, value.view(b, n, num_head, c // num_head), v_shape, v_start,
                                                    ref_windows, lo.view(b, n, -1), None, kidx, num_var)
        sampled2 = sampled.view(-1, c)
        o = torch.addmm(bo, sampled2, wo.t())
        ln = _Ctx()
        y = AddLayerNormFunction.forward(ln, src2, o, gamma, beta, eps)
        box_t, ln_t = _pack(ctx, box, "box"), _pack(ctx, ln, "ln")
        ctx.n_box, ctx.n_ln = len(box_t), len(ln_t)
        ctx.save_for_backward(src2, q2, sampled2, wv, wlo, wo, *box_t, *ln_t)
        ctx.shape, ctx.n_logit_rows = src.shape, wa.shape[0]
        return y.view(src.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        src2, q2, sampled2, wv, wlo, wo = saved[:6]
        box = _unpack(ctx, saved[6:6 + ctx.n_box], "box")
        ln = _unpack(ctx, saved[6 + ctx.n_box:6 + ctx.n_box + ctx.n_ln], "ln")
        c = src2.shape[1]
        dz, _, dgamma, dbeta, _ = AddLayerNormFunction.backward(ln, dy.reshape(-1, c))
        dz2 = dz.view(-1, c)                       # the residual's gradient; the projections accumulate into it below
        g_sampled = dz2.mm(wo)
        gwo = _lin.weight_grad(sampled2, dz2)
        gbo = _lin.column_sum(dz2)
        g_value, _, _, _, g_lo, _, _, _ = _baf.BoxAttnFusedFunction.backward(box, g_sampled.view(ctx.shape))
        g_value2, g_lo2 = g_value.view(-1, c), g_lo.view(-1, g_lo.shape[-1])
        dz2.addmm_(g_value2, wv)                   # d src  += d value . Wv          (in the GEMM: beta = 1)
        dz2.addmm_(g_lo2, wlo)                     # d src  += d [logits | offsets] . [Wa ; Wb]
        gwv = _lin.weight_grad(src2, g_value2)
        gbv = _lin.column_sum(g_value2)
        gwlo = _lin.weight_grad(q2, g_lo2)
        gblo = _lin.column_sum(g_lo2)
        na = ctx.n_logit_rows
        return (dz2.view(ctx.shape), None, None, None, None, None, gwv, gbv, gwlo[:na], gblo[:na], gwlo[na:], gblo[na:],
                gwo, gbo, dgamma, dbeta, None, None, None)


class EncoderFeedForwardHalf(Function):
    @staticmethod
    def forward(ctx, src, w1, b1, w2, b2, gamma, beta, eps):
        c = src.shape[-1]
        src2 = src.reshape(-1, c)
        hidden = torch._addmm_activation(b1, src2, w1.t(), use_gelu=False)   # bias + ReLU in the GEMM epilogue (operators/linear.py)
        o = torch.addmm(b2, hidden, w2.t())
        ln = _Ctx()
        y = AddLayerNormFunction.forward(ln, src2, o, gamma, beta, eps)
        ln_t = _pack(ctx, ln, "ln")
        ctx.save_for_backward(src2, hidden, w1, w2, *ln_t)
        ctx.shape = src.shape
        return y.view(src.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        src2, hidden, w1, w2 = saved[:4]
        ln = _unpack(ctx, saved[4:], "ln")
        c = src2.shape[1]
        dz, _, dgamma, dbeta, _ = AddLayerNormFunction.backward(ln, dy.reshape(-1, c))
        dz2 = dz.view(-1, c)
        g_hidden = dz2.mm(w2)
        gw2 = _lin.weight_grad(hidden, dz2)
        gb2 = _lin.column_sum(dz2)
        masked, gb1 = _lin.relu_backward_column_sum(g_hidden, hidden)   # threshold_backward + bias gradient, one pass
        dz2.addmm_(masked, w1)                     # d src += d hidden . W1          (in the GEMM: beta = 1)
        gw1 = _lin.weight_grad(src2, masked)
        return dz2.view(ctx.shape), gw1, gb1, gw2, gb2, dgamma, dbeta, None


def usable(layer, src, pos, ref_windows):
    """The single-node form covers the training configuration of the path: fp32 on the GPU, gradients on, no dropout, the
    fused sampling kernel applicable, exact-fp32 products (the bf16x3 A/B arm keeps the module-by-module form)."""
    attn = layer.self_attn
    return (_ENABLED and src.is_cuda and src.dtype == torch.float32 and torch.is_grad_enabled() and src.dim() == 3
            and not _lin._ARM_BF16X3 and os.environ.get("EFG_FUSED_LINEAR", "1") != "0"
            and os.environ.get("EFG_FUSED_LN", "1") != "0" and _baf.FUSED_ENABLED
            and not (layer.training and (layer.dropout.p > 0 or layer.dropout1.p > 0 or layer.dropout2.p > 0))
            and attn.head_dim == 32 and attn.num_level * attn.num_point <= 32 and not ref_windows.requires_grad
            and ref_windows.dim() == 3 and src.shape[0] * src.shape[1] >= _lin._FUSED_MIN_ROWS and src.shape[-1] % 4 == 0
            and (pos is None or (pos.shape == src.shape and not pos.requires_grad))   # (backward returns no gradient for pos)
            # what add_layer_norm and the addmm-with-bias products of the two halves assume of the modules
            and all(n.elementwise_affine and n.weight is not None and n.bias is not None and n.normalized_shape[-1] <= 1024
                    for n in (layer.norm1, layer.norm2))
            and all(b is not None for b in (attn.value_proj.bias, attn.linear_attn_bias, attn.linear_box_bias,
                                            attn.out_proj.bias, layer.linear1.bias, layer.linear2.bias))
            and all(p.requires_grad for p in (attn.value_proj.weight, attn.linear_attn_weight, attn.linear_box_weight,
                                              attn.out_proj.weight, layer.linear1.weight, layer.linear2.weight)))


def forward(layer, src, pos, src_shape, src_start_idx, ref_windows):
    attn = layer.self_attn
    src = EncoderAttentionHalf.apply(src, pos, ref_windows, src_shape, src_start_idx, attn.kernel_indices,
                                     attn.value_proj.weight, attn.value_proj.bias, attn.linear_attn_weight,
                                     attn.linear_attn_bias, attn.linear_box_weight, attn.linear_box_bias,
                                     attn.out_proj.weight, attn.out_proj.bias, layer.norm1.weight, layer.norm1.bias,
                                     layer.norm1.eps, attn.num_head, attn.num_variable)
    return EncoderFeedForwardHalf.apply(src, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight,
                                        layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
