"""`efg.operators.ms_deform_attn` on MI355X (mirrors efg/operators/ms_deform_attn.py:24-198)."""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.init import xavier_uniform_

from .box_attention_func import ms_deform_attn_backward, ms_deform_attn_forward


class MSDeformAttnFunction(Function):
    """efg/operators/ms_deform_attn.py:24-52."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        value, sampling_locations, attention_weights = (value.contiguous(), sampling_locations.contiguous(),
                                                        attention_weights.contiguous())
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, attn = ctx.saved_tensors
        grad_value, grad_loc, grad_attn = ms_deform_attn_backward(value, shapes, start, loc, attn,
                                                                  grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_attn, None


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention (Deformable-DETR), drop-in for efg/operators/ms_deform_attn.py:85-198: the
    parameters `sampling_offsets.*`, `attention_weights.*`, `value_proj.*`, `output_proj.*`, their initial values and
    the call `forward(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
    input_padding_mask=None)` are the reference's.

    A query predicts, per head, level and point, a 2-D offset and an attention logit; the value map is sampled
    bilinearly at reference + offset and the samples are mixed with the softmaxed weights.  Sampling + mixing is
    the HIP kernel family of csrc/msda.hip (the same one `BoxAttnFunction` runs); the projections stay on hipBLASLt.
    """

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        head_dim, rem = divmod(d_model, n_heads)
        if rem:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if head_dim & (head_dim - 1):
            warnings.warn("MSDeformAttn: a head width of %d is not a power of two; the sampling kernel vectorises "
                          "best over power-of-two widths" % head_dim)
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.head_dim = head_dim
        self.im2col_step = 64
        samples = n_heads * n_levels * n_points
        self.sampling_offsets = nn.Linear(d_model, 2 * samples)
        self.attention_weights = nn.Linear(d_model, samples)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self.reset_parameters()

    def reset_parameters(self):
        """Offsets start as a star: head h looks along direction 2 pi h / H (scaled to the unit square's edge), its
        p-th point p + 1 steps out, the same on every level; uniform attention; Xavier projections (:107-121)."""
        with torch.no_grad():
            angle = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
            direction = torch.stack((angle.cos(), angle.sin()), dim=-1)
            direction = direction / direction.abs().amax(dim=-1, keepdim=True)              # [H, 2]
            steps = torch.arange(1, self.n_points + 1, dtype=torch.float32)                  # [P]
            star = direction[:, None, None, :] * steps[None, None, :, None]                  # [H, 1, P, 2]
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias.copy_(star.expand(-1, self.n_levels, -1, -1).reshape(-1))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            for proj in (self.value_proj, self.output_proj):
                xavier_uniform_(proj.weight)
                proj.bias.zero_()

    _reset_parameters = reset_parameters  # the reference's name

    def _sampling_locations(self, reference_points, offsets, spatial_shapes):
        """reference_points [N, Lq, L, 2 | 4] in [0, 1]; offsets [N, Lq, H, L, P, 2] -> locations of the same shape.
        2 columns: offsets are in pixels of each level (divided by (W_l, H_l)); 4 columns (cx, cy, w, h): offsets are
        fractions of half the box size, 1 / n_points per unit (:173-187)."""
        ref = reference_points[:, :, None, :, None, :]
        width = reference_points.shape[-1]
        if width == 2:
            wh = spatial_shapes.flip(-1).to(offsets.dtype)  # (H, W) rows -> (W, H)
            return ref + offsets / wh[None, None, None, :, None, :]
        if width == 4:
            return ref[..., :2] + offsets / self.n_points * ref[..., 2:] * 0.5
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(width))

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        n, len_q = query.shape[:2]
        len_in = input_flatten.shape[1]
        if int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) != len_in:
            raise AssertionError("input_spatial_shapes do not add up to the %d flattened input cells" % len_in)
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(n, len_in, self.n_heads, self.head_dim)
        offsets = self.sampling_offsets(query).view(n, len_q, self.n_heads, self.n_levels, self.n_points, 2)
        logits = self.attention_weights(query).view(n, len_q, self.n_heads, self.n_levels * self.n_points)
        weights = F.softmax(logits, dim=-1).view(n, len_q, self.n_heads, self.n_levels, self.n_points)
        locations = self._sampling_locations(reference_points, offsets, input_spatial_shapes)
        sampled = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index, locations, weights,
                                             self.im2col_step)
        return self.output_proj(sampled)
