"""Box3dAttention: the DETR encoder/decoder sampling attention ($CQ/modules/box_attention.py:10-115).

Parameters (`linear_box_weight/bias`, `linear_attn_weight/bias`, `value_proj`, `out_proj`), the
`kernel_indices` buffer, init and forward contract are the reference's; the sampling core is
`BoxAttnFunction` -> csrc/msda.hip.  Dense projections stay on hipBLASLt.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..operators import BoxAttnFunction
from ..operators.linear import Linear, linear
from ..operators import box_attention_func as _baf


class Box3dAttention(nn.Module):
    def __init__(self, d_model, num_level, num_head, with_rotation=True, kernel_size=5):
        super().__init__()
        assert d_model % num_head == 0, "d_model should be divided by num_head"
        num_variable = 5 if with_rotation else 4
        self.im2col_step = 64
        self.d_model, self.num_head, self.num_level = d_model, num_head, num_level
        self.head_dim = d_model // num_head
        self.with_rotation, self.num_variable = with_rotation, num_variable
        self.kernel_size, self.num_point = kernel_size, kernel_size ** 2
        self.linear_box_weight = nn.Parameter(torch.zeros(num_level * num_head * num_variable, d_model))
        self.linear_box_bias = nn.Parameter(torch.zeros(num_head * num_level * num_variable))
        self.linear_attn_weight = nn.Parameter(torch.zeros(num_head * num_level * self.num_point, d_model))
        self.linear_attn_bias = nn.Parameter(torch.zeros(num_head * num_level * self.num_point))
        self.value_proj = Linear(d_model, d_model)
        self.out_proj = Linear(d_model, d_model)
        # k x k lattice in [-0.4, 0.4]^2 (odd k) as (x, y) pairs, :39-50
        if kernel_size % 2 == 0:
            indices = torch.linspace(-kernel_size // 2 + 0.5, kernel_size // 2 - 0.5, kernel_size)
        else:
            indices = torch.linspace(-(kernel_size - 1) // 2, (kernel_size - 1) // 2, kernel_size)
        i, j = torch.meshgrid(indices, indices, indexing="ij")
        self.register_buffer("kernel_indices", torch.stack([j, i], dim=-1).view(-1, 2) / kernel_size)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.constant_(self.out_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.0)
        nn.init.constant_(self.linear_attn_weight, 0.0)
        nn.init.constant_(self.linear_attn_bias, 0.0)
        nn.init.constant_(self.linear_box_weight, 0.0)
        nn.init.uniform_(self.linear_box_bias)

    def _where_to_attend(self, query, v_valid_ratios, ref_windows):
        """:62-95 -> sampling grid [B, L, H, levels, k*k, 2] in normalised (x, y)."""
        B, L = ref_windows.shape[:2]
        offset_boxes = linear(query, self.linear_box_weight, self.linear_box_bias)
        offset_boxes = offset_boxes.view(B, L, self.num_head, self.num_level, self.num_variable)
        ref_windows = ref_windows.unsqueeze(2).unsqueeze(3) if ref_windows.dim() == 3 else ref_windows.unsqueeze(3)
        ref_boxes = ref_windows[..., [0, 1, 3, 4]]
        ref_angles = ref_windows[..., [6]]
        if self.with_rotation:
            offset_boxes, offset_angles = offset_boxes.split(4, dim=-1)
            angles = (ref_angles + offset_angles / 16) * 2 * math.pi
        else:
            angles = ref_angles.expand(B, L, self.num_head, self.num_level, 1)
        boxes = ref_boxes + offset_boxes / 8 * ref_boxes[..., [2, 3, 2, 3]]
        center, size = boxes.unsqueeze(-2).split(2, dim=-1)
        cos_angle, sin_angle = torch.cos(angles), torch.sin(angles)
        rot_matrix = torch.stack([cos_angle, -sin_angle, sin_angle, cos_angle], dim=-1)
        rot_matrix = rot_matrix.view(B, L, self.num_head, self.num_level, 1, 2, 2)
        grid = self.kernel_indices * torch.relu(size)
        grid = center + (grid.unsqueeze(-2) * rot_matrix).sum(-1)
        if v_valid_ratios is not None:
            grid = grid * v_valid_ratios
        return grid.contiguous()

    def forward(self, query, value, v_shape, v_mask, v_start_index, v_valid_ratios, ref_windows):
        B, LQ = query.shape[:2]
        LV = value.shape[1]
        value = self.value_proj(value)
        if v_mask is not None:
            value = value.masked_fill(v_mask[..., None], float(0))
        value = value.view(B, LV, self.num_head, self.head_dim)
        attn_weights = linear(query, self.linear_attn_weight, self.linear_attn_bias)
        if (v_valid_ratios is None and not ref_windows.requires_grad
                and _baf.box_attn_fused_available(value, ref_windows, self.head_dim, self.num_level, self.num_point)):
            # MI355X path: geometry + softmax + sampling in one kernel (csrc/box_fused.hip); same result as the
            # reference sequence below without materialising the [B, LQ, H, L, 25, 2] grid.
            offsets = linear(query, self.linear_box_weight, self.linear_box_bias)
            output = _baf.BoxAttnFusedFunction.apply(value, v_shape, v_start_index, ref_windows, offsets, attn_weights,
                                                     self.kernel_indices, self.num_variable)
            return self.out_proj(output), None
        attn_weights = F.softmax(attn_weights.view(B, LQ, self.num_head, -1), dim=-1)
        attn_weights = attn_weights.view(B, LQ, self.num_head, self.num_level, self.kernel_size, self.kernel_size)
        sampled_grid = self._where_to_attend(query, v_valid_ratios, ref_windows)
        output = BoxAttnFunction.apply(value, v_shape, v_start_index, sampled_grid, attn_weights, self.im2col_step)
        return self.out_proj(output), attn_weights
