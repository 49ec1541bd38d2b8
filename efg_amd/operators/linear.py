"""`Linear`: nn.Linear whose backward takes the bias gradient from libefg_hip.so (csrc/colsum.hip).

The forward `addmm` and the input gradient `grad @ W` are the very calls autograd makes for `F.linear`, so the tuned
hipBLASLt solutions (efg_amd/tuned/gemm_gfx950.csv) keep applying.  Two things change in the backward: the weight
gradient of the 70 688-row encoder layers is computed as 16 row chunks in one batched product (`weight_grad`), and
`grad_output.sum(0)` is replaced: ATen runs the ~100 such reductions of a training step at 1.8 TB/s on the
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


def column_sum(x2, out=None):
    """x2 [rows, cols] fp32 on the GPU (rows may be strided) -> [cols] (into `out`, a contiguous [cols] view, if given),
    deterministic."""
    L.require_gpu(x2)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows, cols = x2.shape
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x2.device)
    ws_bytes = L.lib().efg_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x2.device)
    L.check(L.lib().efg_colsum_f32(x2.data_ptr(), rows, cols, x2.stride(0) if rows > 1 else cols, L.ptr(out), L.ptr(ws),
                                   ws_bytes, L.stream()))
    return out


def relu_backward_column_sum(g2, y):
    """(g2 * (y > 0), its column sums) for contiguous [rows, cols] fp32 GPU matrices: csrc/colsum.hip, one pass."""
    L.require_gpu(g2)
    rows, cols = g2.shape
    masked = torch.empty_like(g2)
    out = torch.empty(cols, dtype=torch.float32, device=g2.device)
    ws_bytes = L.lib().efg_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=g2.device)
    L.check(L.lib().efg_relu_bwd_colsum_f32(L.ptr(g2), L.ptr(y), rows, cols, L.ptr(masked), L.ptr(out), L.ptr(ws), ws_bytes,
                                            L.stream()))
    return masked, out


_SPLIT_MIN_ROWS = 32768
_FUSED_MIN_ROWS = 16384  # linear(): rows from which the custom backward is used
_SPLITS = 16
_WGRAD_ROWMAJOR = True   # (module attribute: the A/B of round 6, profiles/r06o_wgrad_rowmajor.txt, flipped it through an environment switch)
# the decoder-sized Linear + ReLU layers and the self-attention in-projection as fused functions (module attribute: the A/B
# of round 4, profiles/r04_small_fused_ab.txt, flipped it through EFG_SMALL_FUSED; retired as a switch in round 6)
_SMALL_FUSED = True


# The A/B arm of the bench (EFG_GEMM_ARM=bf16x3, never the default): forward and data-gradient products of the long
# matrices, and their weight gradients, as three bf16 MFMA products of split operands (csrc/gemm_bf16x3.hip).  The bias
# gradient and every short matrix stay exact fp32.
_ARM_BF16X3 = os.environ.get("EFG_GEMM_ARM", "") == "bf16x3"
# (No cache of the split weights across calls: one keyed on the parameter's `_version` went stale -- the fused AdamW step
# updates parameters without moving it; tests/test_gemm_bf16x3_gpu.py::test_split_weights_follow_the_optimizer.  The
# forward packs both layouts in one launch and hands the data-gradient one to the backward.)


def _arm_ok(a2, min_cols=64):
    """a2: the [rows, K] operand of a product: the long matrices only (the decoder-sized Linear + ReLU layers also come
    through LinearFunction since round 4 and stay exact fp32).  Below K = 64 the split product loses to fp32 (op bench: 32 -> 256 29.6 vs
    26.2 us; the 32-row weight gradient 44.5 vs 28.5 us)."""
    return (_ARM_BF16X3 and a2.dim() == 2 and a2.stride(1) == 1 and a2.shape[1] % 4 == 0 and a2.stride(0) % 4 == 0
            and a2.data_ptr() % 16 == 0 and a2.shape[0] >= _FUSED_MIN_ROWS and a2.shape[1] >= min_cols)


def weight_grad(x2, g2):
    """grad_outputᵀ @ x -> [out, in].  With tens of thousands of rows and a 256-wide output the product is one
    long reduction over few output tiles; the library's own split runs [256, 70688] x [70688, 256] in 136 us
    (68 TFLOP/s).  Sixteen explicit row chunks as one batched product plus a fixed-order sum of the 16 partial
    matrices take 85 us (scripts/ubench/wgrad_split.py: 333 -> 262 us for the 1024-wide FFN layers, 138 -> 76 us
    for the 200-wide attention logits).  Deterministic, same fp32 products."""
    if _arm_ok(x2) and _arm_ok(g2):
        from . import gemm_bf16x3 as G

        return G.wgrad(g2, x2)
    k = x2.shape[0]
    if k >= _SPLIT_MIN_ROWS and k % _SPLITS == 0 and x2.is_contiguous() and g2.is_contiguous():
        gs = g2.view(_SPLITS, k // _SPLITS, g2.shape[1])
        xs = x2.view(_SPLITS, k // _SPLITS, x2.shape[1])
        return torch.bmm(gs.transpose(1, 2), xs).sum(0)
    if _WGRAD_ROWMAJOR:
        # [out, in] row-major, the parameter's own layout: AccumulateGrad takes the tensor as it is (the transposed view of
        # x2^T . g2 costs a copy launch per parameter and step), and it is the product autograd issues for F.linear on a
        # transposed-view weight (mm_mat2_backward: grad^T . mat1), so the tuned solutions of the plain Linear layers apply
        return g2.t().mm(x2)
    return x2.t().mm(g2).t()


class LinearFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.packed_dgrad = None
        if _arm_ok(x2):
            from . import gemm_bf16x3 as G

            packed_fwd, ctx.packed_dgrad = G.pack_linear_both(weight.detach())
            y = G.gemm(x2, packed_fwd, weight.shape[0], bias=bias, relu=relu)
        elif relu and bias is not None and x2.shape[0] >= _FUSED_MIN_ROWS:
            # bias + ReLU in the GEMM epilogue (hipBLASLt): bit-identical to relu(addmm(...)) on finite values, and the
            # 290 MB activation of the encoder FFN is written once instead of written, read and written again
            # (scripts/ubench/addmm_relu.py: 294 us against 304 + 103 us).  Long matrices only: on the decoder's few
            # thousand rows the epilogue product is no faster than addmm + relu_ (16.5 against 12.9 + 5.5 us, an
            # untuned solution), and its max(x, 0) turns a NaN into 0 where torch.relu keeps it -- the non-finite
            # checks of the engine (tests/test_engine.py) rely on a NaN reaching the loss.
            y = torch._addmm_activation(bias, x2, weight.t(), use_gelu=False)
        else:
            y = torch.addmm(bias, x2, weight.t()) if bias is not None else x2.mm(weight.t())
            if relu:
                y = torch.relu_(y)
        ctx.relu = relu
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        x2, weight, y = ctx.saved_tensors
        g2 = grad.reshape(-1, grad.shape[-1])
        gb = None
        if ctx.relu:
            if (ctx.needs_input_grad[2] and g2.is_contiguous() and y.is_contiguous() and g2.shape[1] % 4 == 0
                    and g2.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0):
                g2, gb = relu_backward_column_sum(g2, y)   # threshold_backward + bias gradient in one pass
            else:
                g2 = torch.ops.aten.threshold_backward(g2, y, 0)  # what autograd runs for relu
        if ctx.needs_input_grad[0] and _arm_ok(g2):
            from . import gemm_bf16x3 as G

            packed = ctx.packed_dgrad if ctx.packed_dgrad is not None else G.pack_linear(weight.detach(), True)
            gx = G.gemm(g2, packed, weight.shape[1]).view(ctx.x_shape)
        else:
            gx = g2.mm(weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        gw = weight_grad(x2, g2) if ctx.needs_input_grad[1] else None
        if gb is None and ctx.needs_input_grad[2]:  # bias=None -> needs_input_grad[2] is False
            gb = column_sum(g2)
        return gx, gw, gb, None


class SelfAttentionInProjFunction(Function):
    """(qk, v) = (qk_in W[:2c]^T + b[:2c], v_in W[2c:]^T + b[2c:]) for nn.MultiheadAttention's packed in_proj parameters
    ($CQ/transformer.py:291-295 calls the module with query = key != value).  Slicing the parameters in Python makes autograd
    build each slice's gradient as a zero-filled [3c, c] tensor plus a copy and add the pieces (17 launches per layer);
    here the backward writes both products and both bias sums straight into ONE [3c, c] / [3c] gradient (6 launches)."""

    @staticmethod
    def forward(ctx, qk_in, v_in, weight, bias):
        c = weight.shape[1]
        q2, v2 = qk_in.reshape(-1, c), v_in.reshape(-1, c)
        qk = torch.addmm(bias[:2 * c], q2, weight[:2 * c].t())
        v = torch.addmm(bias[2 * c:], v2, weight[2 * c:].t())
        ctx.save_for_backward(q2, v2, weight)
        ctx.shapes = (qk_in.shape, v_in.shape)
        return qk.view(*qk_in.shape[:-1], 2 * c), v.view(*v_in.shape[:-1], c)

    @staticmethod
    @once_differentiable
    def backward(ctx, gqk, gv):
        q2, v2, weight = ctx.saved_tensors
        c = weight.shape[1]
        gqk2, gv2 = gqk.reshape(-1, 2 * c), gv.reshape(-1, c)
        if not gqk2.is_contiguous():
            gqk2 = gqk2.contiguous()
        if not gv2.is_contiguous():
            gv2 = gv2.contiguous()
        g_qk_in = gqk2.mm(weight[:2 * c]).view(ctx.shapes[0]) if ctx.needs_input_grad[0] else None
        g_v_in = gv2.mm(weight[2 * c:]).view(ctx.shapes[1]) if ctx.needs_input_grad[1] else None
        gw = gb = None
        if ctx.needs_input_grad[2]:
            gw = torch.empty_like(weight)
            torch.mm(gqk2.t(), q2, out=gw[:2 * c])
            torch.mm(gv2.t(), v2, out=gw[2 * c:])
        if ctx.needs_input_grad[3]:
            gb = torch.empty(3 * c, dtype=torch.float32, device=weight.device)
            column_sum(gqk2, out=gb[:2 * c])
            column_sum(gv2, out=gb[2 * c:])
        return g_qk_in, g_v_in, gw, gb


def self_attention_in_proj(qk_in, v_in, weight, bias):
    """q | k from one projection of `qk_in`, v from `v_in`, with nn.MultiheadAttention's in_proj_weight [3c, c] / in_proj_bias."""
    c = weight.shape[1]
    if (_SMALL_FUSED and qk_in.is_cuda and qk_in.dtype == torch.float32 and torch.is_grad_enabled() and bias is not None
            and (weight.requires_grad or qk_in.requires_grad or v_in.requires_grad)
            and os.environ.get("EFG_FUSED_LINEAR", "1") != "0"):
        return SelfAttentionInProjFunction.apply(qk_in, v_in, weight, bias)
    return F.linear(qk_in, weight[:2 * c], bias[:2 * c]), F.linear(v_in, weight[2 * c:], bias[2 * c:])


def linear(x, weight, bias=None, relu=False):
    """F.linear (followed by ReLU when `relu`); on the GPU, in training, on long matrices with the backward of
    this module."""
    # Only the long matrices (the 70 688-token encoder sequence and the BEV 1x1 convolutions) take the custom
    # backward: there it saves 50-70 us of device time per layer.  On the decoder's few thousand rows the saving is
    # ~5 us per layer while a Python autograd.Function costs ~30 us more host time than F.linear, and the step is
    # close enough to host-bound (~31 ms of launch work against ~36 ms of kernels) for that to matter.
    # Linear + ReLU takes it at every size: bias + ReLU ride in the GEMM epilogue, and the backward's threshold + bias
    # gradient are ONE launch (csrc/colsum.hip) where autograd runs relu, threshold_backward and a two-launch sum.
    if (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
            and (x.numel() >= _FUSED_MIN_ROWS * x.shape[-1] or (_SMALL_FUSED and relu and bias is not None and weight.shape[0] % 4 == 0))
            and (weight.requires_grad or (bias is not None and bias.requires_grad))
            and os.environ.get("EFG_FUSED_LINEAR", "1") != "0"):
        return LinearFunction.apply(x, weight, bias, relu)
    y = F.linear(x, weight, bias)  # short matrices, inference, host tensors (the CPU tests): same math
    return F.relu(y) if relu else y


class Linear(nn.Linear):
    def forward(self, x):
        return linear(x, self.weight, self.bias)
