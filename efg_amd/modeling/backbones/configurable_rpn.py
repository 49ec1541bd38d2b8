"""Dense BEV neck of CenterPoint: the SECOND-style region proposal network (reference:
efg/modeling/backbones/configurable_rpn.py:14-122, same submodule names `blocks.i.j` / `deblocks.i.j`, so reference
checkpoints load by name).

Stage i = ZeroPad2d(1) + 3x3 conv (stride s_i) + `layer_nums[i]` more 3x3 convs, each followed by norm + ReLU; the
output of every stage is brought to one resolution by a transposed conv (stride > 1) or a strided conv (stride < 1)
+ norm + ReLU and the results are concatenated along the channels.  The convolutions are library calls (MIOpen) in
channels-last memory, which is what the sparse backbone's BEV map arrives in."""
import numpy as np
import torch
from torch import nn

from ...operators.batchnorm import run_sequential
from ..common import get_norm


def _cfg(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def _conv_norm_relu(conv, norm, channels):
    return [conv, get_norm(norm, channels), nn.ReLU()]


class RPN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        strides, filters, depths = list(_cfg(cfg, "ds_layer_strides")), list(_cfg(cfg, "ds_num_filters")), list(
            _cfg(cfg, "layer_nums"))
        up_strides, up_filters = list(_cfg(cfg, "us_layer_strides")), list(_cfg(cfg, "us_num_filters"))
        norm = _cfg(cfg, "norm")
        if not (len(strides) == len(filters) == len(depths)) or len(up_strides) != len(up_filters):
            raise ValueError("RPN: ds_* lists (and us_* lists) must have equal lengths")
        self._layer_strides, self._upsample_strides = strides, up_strides
        self.num_channels = sum(up_filters)
        first_up = len(depths) - len(up_strides)  # stages before this one are not upsampled / concatenated
        scale = {u / np.prod(strides[: i + first_up + 1]) for i, u in enumerate(up_strides)}
        if len(scale) > 1:
            raise ValueError("RPN: the upsampled stages do not land on one resolution: %s" % sorted(scale))
        self._first_up = first_up
        blocks, deblocks = [], []
        cin = _cfg(cfg, "num_input_features")
        for i, (s, c, depth) in enumerate(zip(strides, filters, depths)):
            layers = [nn.ZeroPad2d(1)] + _conv_norm_relu(nn.Conv2d(cin, c, 3, stride=s, bias=False), norm, c)
            for _ in range(depth):
                layers += _conv_norm_relu(nn.Conv2d(c, c, 3, padding=1, bias=False), norm, c)
            blocks.append(nn.Sequential(*layers))
            if i >= first_up:
                u, cu = up_strides[i - first_up], up_filters[i - first_up]
                if u > 1:
                    resample = nn.ConvTranspose2d(c, cu, u, stride=u, bias=False)
                else:
                    k = int(np.round(1 / u))
                    resample = nn.Conv2d(c, cu, k, stride=k, bias=False)
                deblocks.append(nn.Sequential(*_conv_norm_relu(resample, norm, cu)))
            cin = c
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)

    @property
    def downsample_factor(self):
        factor = np.prod(self._layer_strides)
        return factor / self._upsample_strides[-1] if self._upsample_strides else factor

    def forward(self, x):
        outs = []
        for i, block in enumerate(self.blocks):
            x = run_sequential(block, x)     # norm + ReLU fused on the GPU (operators/batchnorm.py), same modules
            if i >= self._first_up:
                outs.append(run_sequential(self.deblocks[i - self._first_up], x))
        return torch.cat(outs, dim=1) if outs else x
