"""Sine position embedding over the BEV map ($CQ/modules/position_encoding.py:8-62)."""
import math

import torch
from torch import nn


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def forward(self, x, mask=None):
        if mask is None:
            # a pure function of (shape, dtype, device): computed once, reused every step (the reference
            # rebuilds it with ~15 elementwise kernels over [B, 256, H, W] per step)
            key = (tuple(x.shape[0:1]) + tuple(x.shape[-2:]), x.dtype, x.device, x.is_contiguous(
                memory_format=torch.channels_last))
            cached = self._cache.get(key)
            if cached is None:
                cached = self._compute(x, None)
                if x.dim() == 4 and key[3]:
                    cached = cached.contiguous(memory_format=torch.channels_last)
                self._cache = {key: cached}
            return cached
        return self._compute(x, mask)

    def _compute(self, x, mask=None):
        if mask is not None:
            not_mask = ~mask
            y_embed = not_mask.cumsum(1, dtype=torch.float32)
            x_embed = not_mask.cumsum(2, dtype=torch.float32)
        else:
            h, w = x.shape[-2:]
            y_embed = torch.arange(1, h + 1, dtype=x.dtype, device=x.device)
            x_embed = torch.arange(1, w + 1, dtype=x.dtype, device=x.device)
            y_embed, x_embed = torch.meshgrid(y_embed, x_embed, indexing="ij")
            x_embed = x_embed.unsqueeze(0).repeat(x.shape[0], 1, 1)
            y_embed = y_embed.unsqueeze(0).repeat(x.shape[0], 1, 1)
        if self.normalize:
            eps = 1e-6
            y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        dim_t = self.temperature ** (2 * dim_t.div(2, rounding_mode="floor") / self.num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def build_position_encoding(position_embedding, hidden_dim):
    if position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(hidden_dim // 2, normalize=True)
    raise ValueError(f"not supported {position_embedding}")
