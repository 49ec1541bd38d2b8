"""Dense FPN neck over the BEV maps of the sparse backbone (reference: efg/modeling/backbones/fpn.py).

Same modules / names (`fpn_lateral{2,3,4}`, `fpn_output{2,3,4}`, `bottom_up`), same top-down
arithmetic (:150-169).  Dense Conv2d/BN run on PyTorch-ROCm (MIOpen).

`active_levels` (ours): the levels whose outputs the caller consumes.  ConQueR reads only p3
(config.yaml:116); p2's lateral+3x3 conv at 376^2 (~170 GFLOP/scene), `fpn_output4` and the p5
max-pool feed nothing, receive no gradient in the reference either, and are skipped
when not requested.  The top-down path into a requested level is always evaluated.
"""
import math

import torch.nn.functional as F
from torch import nn

from ..common import Conv2d, c2_xavier_fill, get_norm
from .sparse_net import ShapeSpec, _get, build_sparse_resnet_backbone


class LastLevelMaxPool(nn.Module):
    """P(N+1) = stride-2 subsample of P(N) (fpn.py:186-198)."""

    def __init__(self, in_feature="p5"):
        super().__init__()
        self.num_levels = 1
        self.in_feature = in_feature

    def forward(self, x):
        return [F.max_pool2d(x, kernel_size=1, stride=2, padding=0)]


class FPN(nn.Module):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        self.out_channels = out_channels
        input_shapes = bottom_up.output_shape()
        in_strides = [input_shapes[f].stride for f in in_features]
        in_channels = [input_shapes[f].channels for f in in_features]
        for i, stride in enumerate(in_strides[1:], 1):
            assert stride == 2 * in_strides[i - 1], "Strides {} {} are not log2 contiguous".format(
                stride, in_strides[i - 1])
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channel in enumerate(in_channels):
            lateral_conv = Conv2d(in_channel, out_channels, kernel_size=1, bias=use_bias,
                                  norm=get_norm(norm, out_channels))
            output_conv = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=get_norm(norm, out_channels))
            c2_xavier_fill(lateral_conv)
            c2_xavier_fill(output_conv)
            stage = int(math.log2(in_strides[idx]))
            self.add_module("fpn_lateral{}".format(stage), lateral_conv)
            self.add_module("fpn_output{}".format(stage), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]  # top-down order (low -> high resolution)
        self.output_convs = output_convs[::-1]
        self.top_block = top_block
        self.in_features = list(in_features)
        self.bottom_up = bottom_up
        self._out_feature_strides = {"p{}".format(int(math.log2(s))): s for s in in_strides}
        if self.top_block is not None:
            for s in range(stage, stage + self.top_block.num_levels):
                self._out_feature_strides["p{}".format(s + 1)] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = in_strides[-1]
        assert fuse_type in {"avg", "sum"}
        self._fuse_type = fuse_type
        self.active_levels = None  # None = every level (reference behaviour)

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def set_active_levels(self, levels):
        """Evaluate only what `levels` (e.g. ["p3"]) depend on."""
        self.active_levels = None if levels is None else list(levels)
        if levels is None:
            self.bottom_up.dense_features = None
            return
        lvl_names = ["p{}".format(int(math.log2(s))) for s in
                     [self.bottom_up.output_shape()[f].stride for f in self.in_features]]
        need = [n in levels for n in lvl_names]
        # top-down: level i needs every coarser lateral
        first = min(i for i, n in enumerate(need) if n)
        self.bottom_up.dense_features = self.in_features[first:]

    def forward(self, *args, **kwargs):
        bottom_up_features = self.bottom_up(*args, **kwargs)
        self.__dict__["last_bottom_up"] = bottom_up_features  # (the engine's bucketed gradient exchange hooks these tensors; set past Module.__setattr__)
        levels = ["p{}".format(int(math.log2(self._out_feature_strides[n]))) for n in self._out_features]
        active = set(levels) if self.active_levels is None else set(self.active_levels)
        names_td = [n for n in self._out_features if n in
                    {"p{}".format(int(math.log2(self.bottom_up.output_shape()[f].stride))) for f in self.in_features}]
        names_td = names_td[::-1]  # coarse -> fine, aligned with lateral_convs
        feats = self.in_features[::-1]
        results = {}
        prev = None
        for name, f, lateral_conv, output_conv in zip(names_td, feats, self.lateral_convs, self.output_convs):
            finer_needed = any(n in active for n in names_td[names_td.index(name):])
            if not finer_needed:
                break
            lat = lateral_conv(bottom_up_features[f])
            if prev is not None:
                lat = lat + F.interpolate(prev, scale_factor=2, mode="nearest")
                if self._fuse_type == "avg":
                    lat = lat / 2
            prev = lat
            if name in active:
                results[name] = output_conv(prev)
        if self.top_block is not None:
            top_names = [n for n in self._out_features if n not in names_td]
            if any(n in active for n in top_names):
                src = bottom_up_features.get(self.top_block.in_feature, None)
                if src is None:
                    if self.top_block.in_feature not in results:  # p4 output needed only as the pool's input
                        i = names_td.index(self.top_block.in_feature)
                        raise RuntimeError("top block input %s was not evaluated" % names_td[i])
                    src = results[self.top_block.in_feature]
                for n, r in zip(top_names, self.top_block(src)):
                    results[n] = r
        return {n: results[n] for n in self._out_features if n in results}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


def build_resnet_fpn_backbone(config, input_shape):
    """efg/modeling/backbones/fpn.py:18-37: sparse ResNet bottom-up + FPN(+LastLevelMaxPool)."""
    bottom_up = build_sparse_resnet_backbone(_get(config, "resnet"), input_shape)
    fpn = _get(config, "fpn")
    return FPN(bottom_up=bottom_up, in_features=_get(fpn, "in_features"), out_channels=_get(fpn, "out_channels"),
               norm=_get(fpn, "norm"), top_block=LastLevelMaxPool(in_feature=_get(fpn, "top_block_in_feature")),
               fuse_type=_get(fpn, "fuse_type"))
