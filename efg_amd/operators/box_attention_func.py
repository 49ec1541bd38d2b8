"""`efg.operators.box_attention_func` on MI355X (mirrors efg/operators/box_attention_func.py:9-64).

`box_attn_forward/backward` and `ms_deform_attn_forward/backward` keep the Python-visible
signatures of the reference bindings (efg/operators/src/box_attn/box_attn.h:29-83,
efg/operators/src/deform_attn/ms_deform_attn.h:22-63); both names run the same HIP kernel family
(csrc/msda.hip).  fp32 only, like the ConQueR path (`custom_fwd(cast_inputs=torch.float32)`).
"""
import math
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from .. import _prof

_CHECK_BINS = os.environ.get("EFG_CHECK_BINS", "0") == "1"


def _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    L.require_gpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    for name, t in (("value", value), ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError(name + " must be contiguous.")  # CHECK_INPUT, efg_cutils.h:12-15
        if t.dtype != torch.float32:
            raise RuntimeError(name + " must be float32")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    batch = value.size(0)
    step = min(batch, im2col_step)
    if batch > 0 and batch % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))  # box_attn.cu:39-41


def box_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value[B,S,H,D], shapes i64[L,2], start i64[L], loc[B,Lq,H,L,P,2], attn[B,Lq,H,L,P(or k,k)] -> [B,Lq,H*D]."""
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    b, s, h, d = value.shape
    lq, l, p = sampling_loc.size(1), spatial_shapes.size(0), sampling_loc.size(4)
    out = torch.empty((b, lq, h * d), dtype=value.dtype, device=value.device)
    # algorithmic bytes: value once + loc/attn + out; flops ~ 10 per (sample, channel)
    cost = lambda: (4 * (b * s * h * d + b * lq * h * l * p * 3 + b * lq * h * d), 10 * b * lq * h * l * p * d)  # noqa: E731
    with _prof.timed("msda_kernel<false>", cost):
        L.check(L.lib().efg_msda_forward_f32(L.ptr(value), L.ptr(spatial_shapes.contiguous()),
                                             L.ptr(level_start_index.contiguous()), L.ptr(sampling_loc),
                                             L.ptr(attn_weight), b, s, h, d, l, lq, p, L.ptr(out), L.stream()))
    return out


def box_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]"""
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    L.require_gpu(grad_output)
    grad_output = grad_output.contiguous()
    b, s, h, d = value.shape
    lq, l, p = sampling_loc.size(1), spatial_shapes.size(0), sampling_loc.size(4)
    grad_value = torch.zeros_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    # bwd algorithmic bytes: fwd reads + grad_out, writes grad_value + grad_loc + grad_attn
    cost = lambda: (4 * (2 * b * s * h * d + 2 * b * lq * h * l * p * 3 + 2 * b * lq * h * d), 30 * b * lq * h * l * p * d)  # noqa: E731
    grid = (l == 1 and d == 32 and s == lq and s >= 1024)
    with _prof.timed("msda_bwd_grid_kernel<32>" if grid else "msda_kernel<true>", cost):
        L.check(L.lib().efg_msda_backward_f32(L.ptr(value), L.ptr(spatial_shapes.contiguous()),
                                              L.ptr(level_start_index.contiguous()), L.ptr(sampling_loc),
                                              L.ptr(attn_weight), L.ptr(grad_output), b, s, h, d, l, lq, p,
                                              L.ptr(grad_value), L.ptr(grad_loc), L.ptr(grad_attn), L.stream()))
    return [grad_value, grad_loc, grad_attn]


ms_deform_attn_forward = box_attn_forward
ms_deform_attn_backward = box_attn_backward


class BoxAttnFunction(Function):
    """efg/operators/box_attention_func.py:9-64 (inputs are cast to fp32 like custom_fwd there)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        value, sampling_locations, attention_weights = (value.float().contiguous(),
                                                        sampling_locations.float().contiguous(),
                                                        attention_weights.float().contiguous())
        output = box_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                  attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, attn = ctx.saved_tensors
        grad_value, grad_loc, grad_attn = box_attn_backward(value, shapes, start, loc, attn,
                                                            grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_attn, None


# ---- box geometry: where a query's k x k lattice lands on the value map --------------------------------------
def box_sampling_grid(ref_windows, offsets, kernel_indices, num_head, num_level, with_rotation, valid_ratios=None):
    """The sampling geometry of Box3dAttention (contract: SURVEY.md B.5; reference $CQ/modules/box_attention.py:62-95;
    the same arithmetic, in the same order, as csrc/box_fused.hip evaluates per lane).

    ref_windows [B, Lq, 7] (x, y, z, l, w, h, angle/2pi), normalised; offsets [B, Lq, H * L * V] raw outputs of the
    box Linear with V = 5 (dx, dy, dl, dw, dangle) when `with_rotation` else 4; kernel_indices [P, 2] lattice.
    Returns the grid [B, Lq, H, L, P, 2] of normalised (x, y) sampling locations:
        centre = (x, y) + (dx, dy) / 8 * (l, w);  size = (l, w) + (dl, dw) / 8 * (l, w)
        theta  = (angle + dangle / 16) * 2 pi
        point  = centre + R(theta) * (lattice * relu(size))
    Plain slicing only -- no index lists, so it is legal inside a HIP-graph capture."""
    b, lq = ref_windows.shape[:2]
    nvar = 5 if with_rotation else 4
    off = offsets.reshape(b, lq, num_head, num_level, nvar)
    ref = ref_windows.reshape(b, lq, 1, 1, ref_windows.shape[-1]) if ref_windows.dim() == 3 else ref_windows.unsqueeze(3)
    cx0, cy0, length, width, angle = ref[..., 0], ref[..., 1], ref[..., 3], ref[..., 4], ref[..., 6]
    cx = cx0 + off[..., 0] / 8 * length
    cy = cy0 + off[..., 1] / 8 * width
    sx = torch.relu(length + off[..., 2] / 8 * length)
    sy = torch.relu(width + off[..., 3] / 8 * width)
    theta = ((angle + off[..., 4] / 16) * 2 * math.pi) if with_rotation else angle.expand(b, lq, num_head, num_level)
    cos_t, sin_t = torch.cos(theta).unsqueeze(-1), torch.sin(theta).unsqueeze(-1)
    kx = kernel_indices[:, 0] * sx.unsqueeze(-1)  # [B, Lq, H, L, P]
    ky = kernel_indices[:, 1] * sy.unsqueeze(-1)
    gx = cx.unsqueeze(-1) + (kx * cos_t + ky * (-sin_t))
    gy = cy.unsqueeze(-1) + (kx * sin_t + ky * cos_t)
    grid = torch.stack((gx, gy), dim=-1)
    if valid_ratios is not None:
        grid = grid * valid_ratios
    return grid.contiguous()


# ---- fused Box3dAttention sampling (csrc/box_fused.hip) -------------------------------------------------
FUSED_ENABLED = True


def box_attn_fused_available(value, ref_windows, head_dim, num_level, num_point):
    return (FUSED_ENABLED and value.is_cuda and head_dim == 32 and ref_windows.dim() == 3
            and num_level * num_point <= 128 and value.dtype == torch.float32)


class BoxAttnFusedFunction(Function):
    """(value[B,S,H,D], shapes, start, ref[B,Lq,7], offsets[B,Lq,H*L*V], logits[B,Lq,H*L*P], kernel_indices[P,2])
    -> [B,Lq,H*D].  Equivalent to _where_to_attend + softmax + BoxAttnFunction of the reference module."""

    @staticmethod
    def forward(ctx, value, shapes, start, ref, offsets, logits, kidx, num_var):
        """`logits` None: `offsets` is the output [B, Lq, H*L*P + H*L*V] of ONE projection (Box3dAttention concatenates
        its two weight matrices), logits in the leading H*L*P columns, box offsets behind them; the kernels read and
        write both through the shared row stride, and the backward returns ONE gradient matrix (one data-gradient
        product and one weight-gradient product instead of two of each and an addition of their input gradients)."""
        shared = logits is None
        if shared:
            L.require_gpu(value, ref, offsets, kidx)
            value, ref, offsets, kidx = (t.float().contiguous() for t in (value, ref, offsets, kidx))
        else:
            L.require_gpu(value, ref, offsets, logits, kidx)
            value, ref, offsets, logits, kidx = (t.float().contiguous() for t in (value, ref, offsets, logits, kidx))
        b, s, h, d = value.shape
        lq, l, p = ref.shape[1], shapes.size(0), kidx.shape[0]
        n_lg = h * l * p
        if shared and offsets.shape[-1] != n_lg + h * l * num_var:
            raise ValueError("box_attn_fused: shared projection of width %d, expected %d logits + %d offsets" % (
                offsets.shape[-1], n_lg, h * l * num_var))
        out = torch.empty((b, lq, h * d), dtype=torch.float32, device=value.device)
        cost = lambda: (4 * (b * s * h * d + b * lq * (7 + h * l * (num_var + p)) + b * lq * h * d),  # noqa: E731
                        10 * b * lq * h * l * p * d)
        rs = offsets.shape[-1] if shared else 0
        off_ptr = offsets.data_ptr() + 4 * n_lg if shared else L.ptr(offsets)
        lg_ptr = offsets.data_ptr() if shared else L.ptr(logits)
        with _prof.timed("box_fwd_kernel", cost):
            L.check(L.lib().efg_box_attn_fused_forward_strided_f32(L.ptr(value), L.ptr(shapes.contiguous()),
                                                                   L.ptr(start.contiguous()), L.ptr(ref), off_ptr, rs,
                                                                   lg_ptr, rs, L.ptr(kidx), b, s, h, d, l, lq, p, num_var,
                                                                   L.ptr(out), L.stream()))
        ctx.save_for_backward(value, shapes, start, ref, offsets, logits, kidx)
        ctx.num_var = num_var
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, ref, offsets, logits, kidx = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        b, s, h, d = value.shape
        lq, l, p = ref.shape[1], shapes.size(0), kidx.shape[0]
        shared = logits is None
        n_lg = h * l * p
        grad_value = torch.zeros_like(value)
        grad_off = torch.empty_like(offsets)     # (shared projection: the gradient of the whole [.., logits | offsets] matrix)
        grad_logits = None if shared else torch.empty_like(logits)
        rs = offsets.shape[-1] if shared else 0
        off_ptr = offsets.data_ptr() + 4 * n_lg if shared else L.ptr(offsets)
        lg_ptr = offsets.data_ptr() if shared else L.ptr(logits)
        goff_ptr = grad_off.data_ptr() + 4 * n_lg if shared else L.ptr(grad_off)
        glg_ptr = grad_off.data_ptr() if shared else L.ptr(grad_logits)
        cost = lambda: (4 * (2 * b * s * h * d + 2 * b * lq * (h * l * (ctx.num_var + p)) + 2 * b * lq * h * d),  # noqa: E731
                        30 * b * lq * h * l * p * d)
        grid = (l == 1 and s == lq and s >= 1024)
        name = ("box_bwd_tile_kernel" if l * p <= 32 else "box_bwd_kernel<32, true, 128>") if grid \
            else "box_bwd_kernel<32, false, 128>"
        ws, ws_bytes = None, 0
        if not grid or l * p <= 32:
            # scratch for the binned grad_value reduction (csrc/box_fused.hip): every corner of the free-position
            # (decoder) queries; for the encoder's tile kernel the corners that leave the window of their query tile
            ws_bytes = L.lib().efg_box_attn_fused_backward_workspace_bytes(b, s, h, l, lq, p)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
        with _prof.timed(name, cost):
            L.check(L.lib().efg_box_attn_fused_backward_strided_f32(L.ptr(value), L.ptr(shapes.contiguous()),
                                                                    L.ptr(start.contiguous()), L.ptr(ref), off_ptr, rs,
                                                                    lg_ptr, rs, L.ptr(kidx), L.ptr(grad_output), b, s, h,
                                                                    d, l, lq, p, ctx.num_var, L.ptr(grad_value),
                                                                    goff_ptr, glg_ptr, L.ptr(ws), ws_bytes, L.stream()))
        if ws is not None and _CHECK_BINS:
            # first word of the scratch: entries that did not fit the bin box_bin_count_kernel sized for them (a
            # disagreement between the counting and the writing kernel).  Must be 0; reading it is a sync, so only
            # the tests (EFG_CHECK_BINS=1) do.
            dropped = int(ws[:4].view(torch.int32).item())
            if dropped:
                raise RuntimeError("box_attn_fused backward: %d binned grad_value entries overflowed their bin" % dropped)
        return grad_value, None, None, None, grad_off, grad_logits, None, None
