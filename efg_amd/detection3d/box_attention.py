"""Box3dAttention: the sampling attention of the Voxel-DETR encoder (axis-aligned windows) and decoder (rotated
boxes).  Drop-in for $CQ/modules/box_attention.py:10-115: same parameter names and shapes (`linear_box_weight/bias`,
`linear_attn_weight/bias`, `value_proj.*`, `out_proj.*`), the `kernel_indices` buffer, the same initial values and the
same forward contract `(query, value, v_shape, v_mask, v_start_index, v_valid_ratios, ref_windows) -> (out, weights)`.

MI355X structure: a query contributes three small Linears (value is projected once per call); everything between
them and the output projection -- box geometry, softmax over the L * k * k logits, bilinear sampling -- is ONE HIP
kernel (csrc/box_fused.hip) that never writes the [B, Lq, H, L, k*k, 2] grid or the softmaxed weights to HBM.  When
that kernel does not apply (head width != 32, per-level valid ratios, a gradient requested for the windows) the same
quantities are materialised with `box_sampling_grid` and handed to the plain sampling op (`BoxAttnFunction`,
csrc/msda.hip), which is what the reference module does on every call.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..operators import BoxAttnFunction
from ..operators import box_attention_func as _baf
from ..operators.linear import Linear, linear


# (False: logits and offsets from two projections, as the reference module computes them -- the A/B of round 5)
_SHARED_PROJECTION = True   # (module attribute: the A/B of round 5 and the tests flip it in-process)


def _lattice(k):
    """k x k sampling lattice as (x, y) pairs, y-major, spanning (-0.5, 0.5) exclusive: {(j, i) - (k-1)/2} / k."""
    axis = (torch.arange(k, dtype=torch.float32) - (k - 1) / 2) / k
    return torch.stack((axis.repeat(k), axis.repeat_interleave(k)), dim=-1)


class Box3dAttention(nn.Module):
    def __init__(self, d_model, num_level, num_head, with_rotation=True, kernel_size=5):
        super().__init__()
        if d_model % num_head:
            raise ValueError("d_model (%d) must be a multiple of num_head (%d)" % (d_model, num_head))
        self.d_model, self.num_head, self.num_level = d_model, num_head, num_level
        self.head_dim = d_model // num_head
        self.with_rotation = with_rotation
        self.num_variable = 5 if with_rotation else 4          # dx, dy, dl, dw (, dangle) per head and level
        self.kernel_size, self.num_point = kernel_size, kernel_size * kernel_size
        self.im2col_step = 64
        per_query = num_head * num_level
        self.linear_box_weight = nn.Parameter(torch.empty(per_query * self.num_variable, d_model))
        self.linear_box_bias = nn.Parameter(torch.empty(per_query * self.num_variable))
        self.linear_attn_weight = nn.Parameter(torch.empty(per_query * self.num_point, d_model))
        self.linear_attn_bias = nn.Parameter(torch.empty(per_query * self.num_point))
        self.value_proj = Linear(d_model, d_model)
        self.out_proj = Linear(d_model, d_model)
        self.register_buffer("kernel_indices", _lattice(kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        """Start as a plain average over a window of the reference size: zero box / attention weights, box bias
        U(0, 1) (reference :52-60), Xavier projections."""
        with torch.no_grad():
            for p in (self.linear_box_weight, self.linear_attn_weight, self.linear_attn_bias):
                p.zero_()
            self.linear_box_bias.uniform_()
            for proj in (self.value_proj, self.out_proj):
                nn.init.xavier_uniform_(proj.weight)
                proj.bias.zero_()

    _reset_parameters = reset_parameters  # the reference's name

    def _where_to_attend(self, query, v_valid_ratios, ref_windows):
        """Sampling grid [B, Lq, H, L, k*k, 2] for `query` (the reference's method name, :62-95)."""
        offsets = linear(query, self.linear_box_weight, self.linear_box_bias)
        return _baf.box_sampling_grid(ref_windows, offsets, self.kernel_indices, self.num_head, self.num_level,
                                      self.with_rotation, v_valid_ratios)

    def forward(self, query, value, v_shape, v_mask, v_start_index, v_valid_ratios, ref_windows):
        b, lq = query.shape[:2]
        value = self.value_proj(value)
        if v_mask is not None:
            value = value.masked_fill(v_mask[..., None], 0.0)
        value = value.view(b, value.shape[1], self.num_head, self.head_dim)
        fused = (v_valid_ratios is None and not ref_windows.requires_grad and
                 _baf.box_attn_fused_available(value, ref_windows, self.head_dim, self.num_level, self.num_point))
        if fused and _SHARED_PROJECTION:
            # ONE projection of the query for the attention logits and the box offsets (the two weight matrices stacked:
            # two tiny copies), handed to the sampling kernel as two column ranges of one matrix: one product instead of
            # two in the forward, one data-gradient and one weight-gradient product instead of two of each in the
            # backward, and the two input gradients -- [B, Lq, 256] each, 72 MB on the encoder's 70 688 tokens -- are
            # never added because they are never separate.  Parameters, names and values are the reference's.
            lo = linear(query, torch.cat((self.linear_attn_weight, self.linear_box_weight), 0),
                        torch.cat((self.linear_attn_bias, self.linear_box_bias), 0))   # [B, Lq, H*L*k*k + H*L*V]
            sampled = _baf.BoxAttnFusedFunction.apply(value, v_shape, v_start_index, ref_windows, lo, None,
                                                      self.kernel_indices, self.num_variable)
            return self.out_proj(sampled), None
        logits = linear(query, self.linear_attn_weight, self.linear_attn_bias)   # [B, Lq, H * L * k*k]
        offsets = linear(query, self.linear_box_weight, self.linear_box_bias)    # [B, Lq, H * L * V]
        if fused:
            sampled = _baf.BoxAttnFusedFunction.apply(value, v_shape, v_start_index, ref_windows, offsets, logits,
                                                      self.kernel_indices, self.num_variable)
            return self.out_proj(sampled), None
        weights = F.softmax(logits.view(b, lq, self.num_head, self.num_level * self.num_point), dim=-1)
        weights = weights.view(b, lq, self.num_head, self.num_level, self.kernel_size, self.kernel_size)
        grid = _baf.box_sampling_grid(ref_windows, offsets, self.kernel_indices, self.num_head, self.num_level,
                                      self.with_rotation, v_valid_ratios)
        sampled = BoxAttnFunction.apply(value, v_shape, v_start_index, grid, weights, self.im2col_step)
        return self.out_proj(sampled), weights
