"""Sine position embedding over the BEV map (reference: $CQ/modules/position_encoding.py:8-62, same values).

Without a padding mask the embedding is separable: channel block 0 depends only on the row, block 1 only on the
column.  It is therefore built from two 1-D tables ([H, F] and [W, F]) that are broadcast into the [B, 2F, H, W]
map once per (shape, dtype, device, layout) and cached -- the reference rebuilds the full map with ~15 elementwise
kernels every step.  With a mask the coordinates are cumulative sums of the valid cells and the general path runs.
"""
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

    # ---- pieces ---------------------------------------------------------------------------------------------
    def _periods(self, device):
        """temperature ** (2 * (c // 2) / F) for c = 0 .. F-1: channels 2i and 2i+1 share a period."""
        c = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
        return self.temperature ** (2 * c.div(2, rounding_mode="floor") / self.num_pos_feats)

    def _wave(self, coord, periods):
        """coord [...] -> [..., F]: sin on even channels, cos on odd ones."""
        phase = coord.unsqueeze(-1) / periods
        even = (torch.arange(self.num_pos_feats, device=coord.device) % 2 == 0)
        return torch.where(even, phase.sin(), phase.cos())

    def _normalized(self, coord, last):
        """(coord - 0.5) / (last + eps) * scale: cell centres mapped to (0, scale)."""
        return (coord - 0.5) / (last + 1e-6) * self.scale if self.normalize else coord

    def _separable(self, batch, h, w, dtype, device):
        periods = self._periods(device)
        rows = torch.arange(1, h + 1, dtype=dtype, device=device)
        cols = torch.arange(1, w + 1, dtype=dtype, device=device)
        table_y = self._wave(self._normalized(rows, rows[-1]), periods)  # [H, F]
        table_x = self._wave(self._normalized(cols, cols[-1]), periods)  # [W, F]
        f = self.num_pos_feats
        out = torch.empty(batch, 2 * f, h, w, dtype=table_y.dtype, device=device)
        out[:, :f] = table_y.t()[None, :, :, None]
        out[:, f:] = table_x.t()[None, :, None, :]
        return out

    def _masked(self, mask):
        valid = ~mask
        y = valid.cumsum(1, dtype=torch.float32)
        x = valid.cumsum(2, dtype=torch.float32)
        periods = self._periods(mask.device)
        wave_y = self._wave(self._normalized(y, y[:, -1:, :]), periods)
        wave_x = self._wave(self._normalized(x, x[:, :, -1:]), periods)
        return torch.cat((wave_y, wave_x), dim=3).permute(0, 3, 1, 2)

    # ---- module ---------------------------------------------------------------------------------------------
    def forward(self, x, mask=None):
        if mask is not None:
            return self._masked(mask)
        channels_last = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
        key = (x.shape[0], x.shape[-2], x.shape[-1], x.dtype, x.device, channels_last)
        hit = self._cache.get(key)
        if hit is None:
            hit = self._separable(x.shape[0], x.shape[-2], x.shape[-1], x.dtype, x.device)
            if channels_last:
                hit = hit.contiguous(memory_format=torch.channels_last)
            self._cache = {key: hit}  # one shape at a time: the BEV map of the running configuration
        return hit


def build_position_encoding(position_embedding, hidden_dim):
    if position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(hidden_dim // 2, normalize=True)
    raise ValueError(f"not supported {position_embedding}")
