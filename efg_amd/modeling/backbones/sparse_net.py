"""Sparse 3-D backbones over efg_amd.spconv (reference: efg/modeling/backbones/sparse_net.py).

Module / parameter names, channel plans, strides, kernel geometry and the dense BEV output layout
follow the reference so its state dicts load unchanged:
  SparseBasicStem :79-103, SparseBasicResBlock :120-165, SparseResNet :239-316,
  build_sparse_resnet_backbone :318-397, SparseBasicBlock :429-470, SpMiddleResNetFHD :473-545.
"""
import collections

import numpy as np
from torch import nn

from .. import common
from ... import spconv
from ...operators.batchnorm import bn_act
from ...spconv import SparseConv3d, SubMConv3d

ShapeSpec = collections.namedtuple("ShapeSpec", ["channels", "height", "width", "stride"],
                                   defaults=(None, None, None, None))


class SparseBasicStem(spconv.SparseModule):
    """SparseConv3d(k3,s2,p1) + 2 x SubMConv3d(k3), each followed by norm + activation (:85-95)."""

    def __init__(self, in_channels=16, out_channels=32, stem_width=32, norm="BN1d", activation=None, indice_key=None):
        super().__init__()
        self.out_channels = out_channels
        self.conv1 = spconv.SparseSequential(
            SparseConv3d(in_channels, stem_width, 3, 2, padding=1, bias=False),
            common.get_norm(norm, stem_width),
            common.get_activation(activation),
            SubMConv3d(stem_width, stem_width, 3, padding=1, bias=False, indice_key=indice_key),
            common.get_norm(norm, stem_width),
            common.get_activation(activation),
            SubMConv3d(stem_width, out_channels, 3, padding=1, bias=False, indice_key=indice_key),
            common.get_norm(norm, out_channels),
            common.get_activation(activation),
        )

    def forward(self, x):
        return self.conv1(x)

    @property
    def stride(self):
        return 2


class SparseBasicResBlock(spconv.SparseModule):
    """Two 3x3x3 convs + identity / strided-conv shortcut (:120-165).  With stride 2 both the first
    conv and the shortcut are regular SparseConv3d over the same input -> row-aligned outputs."""

    def __init__(self, in_channels=32, out_channels=64, stride=1, norm="BN1d", activation=None, indice_key=None):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        if in_channels != out_channels:
            self.shortcut = spconv.SparseSequential(
                SparseConv3d(in_channels, out_channels, 3, padding=1, stride=stride, bias=False),
                common.get_norm(norm, out_channels),
            )
        else:
            self.shortcut = None
        self.activation = common.get_activation(activation)
        first = (SubMConv3d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False,
                            indice_key=indice_key) if stride == 1 else
                 SparseConv3d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False))
        self.conv = spconv.SparseSequential(
            first,
            common.get_norm(norm, out_channels),
            common.get_activation(activation),
            SubMConv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                       indice_key=indice_key),
            common.get_norm(norm, out_channels),
        )

    def forward(self, x):
        mods = [m for m in self.conv._modules.values() if m is not None]
        last = mods[-1]
        if type(self.activation) is nn.ReLU and isinstance(last, nn.BatchNorm1d):
            # relu(bn2(conv2(.)) + shortcut) as one fused op (operators/batchnorm.py) when the features are on the GPU
            pair = None
            if self.shortcut is not None and self.stride != 1 and len(mods) == 5 and type(mods[2]) is nn.ReLU:
                # the strided first convolution and the strided shortcut read the same input through the same rulebook: both
                # products, their joint data gradient and both weight gradients as ONE launch each (spconv.conv_pair_bn_act)
                sc = [m for m in self.shortcut._modules.values() if m is not None]
                if len(sc) == 2:
                    pair = spconv.conv_pair_bn_act(mods[0], mods[1], True, sc[0], sc[1], False, x)
            if pair is not None:
                pre, shortcut = pair
            else:
                shortcut = self.shortcut(x) if self.shortcut is not None else x
                pre = spconv.run_modules(mods[:-2], x)
            # last convolution + norm + residual + ReLU as one autograd node where that applies (spconv.conv_bn_act)
            fused = spconv.conv_bn_act(mods[-2], pre, last, relu=True, residual=shortcut.features)
            if fused is not None:
                return fused
            out = spconv.run_modules(mods[-2:-1], pre)
            return out.replace_feature(bn_act(out.features, last, relu=True, residual=shortcut.features))
        out = self.conv(x)
        shortcut = self.shortcut(x) if self.shortcut is not None else x
        out = out.replace_feature(out.features + shortcut.features)
        return out.replace_feature(self.activation(out.features))


class SparseBottleneckBlock(spconv.SparseModule):
    """1x1x1 SubM -> 3x3x3 (SubM, or strided SparseConv3d in a stage's first block) -> 1x1x1 SubM, each + norm, with the
    identity / strided 3x3x3-conv shortcut of the basic block (reference :168-237; depth 50 of
    build_sparse_resnet_backbone :388).  Same module tree -- `shortcut.{0,1}`, `conv.{0..7}` -- so its state dicts load.

    One deliberate difference: the reference sizes the LAST norm with `bottleneck_channels` (:214) although it
    normalises the `out_channels`-wide output of the second 1x1x1 convolution, which only runs when the two widths are
    equal; here it is sized with `out_channels` (identical parameters in the case that runs, a working block otherwise).
    The 1x1x1 convolutions share the block's `indice_key` with the 3x3x3 one as in the reference; efg_amd.spconv keys its
    rulebook cache by (key, kernel size), so they never see each other's tables."""

    def __init__(self, in_channels, out_channels, bottleneck_channels, stride=1, norm="BN1d", activation=None, indice_key=None):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        if in_channels != out_channels:
            self.shortcut = spconv.SparseSequential(
                SparseConv3d(in_channels, out_channels, 3, padding=1, stride=stride, bias=False),
                common.get_norm(norm, out_channels),
            )
        else:
            self.shortcut = None
        self.activation = common.get_activation(activation)
        mid = (SubMConv3d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=1, padding=1, bias=False,
                          indice_key=indice_key) if stride == 1 else
               SparseConv3d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride, padding=1, bias=False))
        self.conv = spconv.SparseSequential(
            SubMConv3d(in_channels, bottleneck_channels, kernel_size=1, stride=1, bias=False, indice_key=indice_key),
            common.get_norm(norm, bottleneck_channels),
            common.get_activation(activation),
            mid,
            common.get_norm(norm, bottleneck_channels),
            common.get_activation(activation),
            SubMConv3d(bottleneck_channels, out_channels, kernel_size=1, bias=False, indice_key=indice_key),
            common.get_norm(norm, out_channels),
        )

    def forward(self, x):
        mods = [m for m in self.conv._modules.values() if m is not None]
        last = mods[-1]
        if type(self.activation) is nn.ReLU and isinstance(last, nn.BatchNorm1d):
            # relu(bn3(conv3(.)) + shortcut) as one fused op (operators/batchnorm.py), as in the basic block
            shortcut = self.shortcut(x) if self.shortcut is not None else x
            pre = spconv.run_modules(mods[:-2], x)
            # last convolution + norm + residual + ReLU as one autograd node where that applies (spconv.conv_bn_act)
            fused = spconv.conv_bn_act(mods[-2], pre, last, relu=True, residual=shortcut.features)
            if fused is not None:
                return fused
            out = spconv.run_modules(mods[-2:-1], pre)
            return out.replace_feature(bn_act(out.features, last, relu=True, residual=shortcut.features))
        out = self.conv(x)
        shortcut = self.shortcut(x) if self.shortcut is not None else x
        out = out.replace_feature(out.features + shortcut.features)
        return out.replace_feature(self.activation(out.features))


def make_stage(block_class, num_blocks, first_stride, **kwargs):
    blocks = []
    for i in range(num_blocks):
        blocks.append(block_class(stride=first_stride if i == 0 else 1, **kwargs))
        kwargs["in_channels"] = kwargs["out_channels"]
    return blocks


class SparseResNet(nn.Module):
    """stem -> res2..resN; each requested feature gets a `<name>_out` head that collapses z with
    SparseConv3d((3,1,1),(2,1,1),pad (1,0,0)) + norm + ReLU and is returned DENSE as
    [B, C*D, H, W] (:273-282, :302-307).

    `dense_features` (ours): restrict which `<name>_out` heads are actually evaluated.  The
    ConQueR graph consumes only p3 (config.yaml:116), so `res2_out` (+ its 217 MB/scene dense
    tensor) feeds nothing; skipping it changes no output and no gradient (SURVEY.md §7 "dead
    compute").  Parameters stay in the state dict either way.
    """

    def __init__(self, stem, stages, out_channels=None, out_features=None, norm=None):
        super().__init__()
        self.stem = stem
        self.out_channels = out_channels
        current_stride = self.stem.stride
        self._out_feature_strides = {"stem": current_stride}
        self._out_feature_channels = {"stem": self.stem.out_channels}
        self.stages_and_names = []
        for i, blocks in enumerate(stages):
            stage = spconv.SparseSequential(*blocks)
            name = "res" + str(i + 2)
            self.add_module(name, stage)
            self.stages_and_names.append((stage, name))
            self._out_feature_strides[name] = current_stride = int(current_stride * np.prod([k.stride for k in blocks]))
            self._out_feature_channels[name] = blocks[-1].out_channels
        if out_features is None:
            out_features = [name]
        self._out_features = list(out_features)
        children = [x[0] for x in self.named_children()]
        for out_feature in self._out_features:
            assert out_feature in children, "Available children: {}".format(", ".join(children))
        out_channels_multiplier = [6, 3, 2]  # D after the z-collapsing head at 41 -> 21 -> 11 -> 6 -> 3 (:273)
        for idx, out_feature in enumerate(self._out_features):
            channels = self._out_feature_channels[out_feature]
            out_layer = spconv.SparseSequential(
                SparseConv3d(channels, channels, (3, 1, 1), (2, 1, 1), padding=(1, 0, 0), bias=False),
                common.get_norm(norm, channels),
                nn.ReLU(),
            )
            self.add_module(out_feature + "_out", out_layer)
            self._out_feature_channels[out_feature] *= out_channels_multiplier[idx]
        self.dense_features = None  # None = all of out_features

    def forward(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]  # (z+1, y, x), :285
        coors = coors.int()
        x = spconv.SparseConvTensor(voxel_features, coors, sparse_shape.tolist(), batch_size)
        wanted = self._out_features if self.dense_features is None else [
            f for f in self._out_features if f in self.dense_features]
        outputs = {}
        last = max((i for i, (_, n) in enumerate(self.stages_and_names) if n in wanted), default=-1)
        self._prefetch_geometry(x, wanted, last)
        x = self.stem(x)
        if "stem" in wanted:
            outputs["stem"] = x
        for i, (stage, name) in enumerate(self.stages_and_names):
            if i > last:
                break
            x = stage(x)
            if name in wanted:
                outputs[name] = x
        for out_feature in wanted:
            out = getattr(self, out_feature + "_out")(outputs[out_feature])
            outputs[out_feature] = out.dense_bev()  # == out.dense().view(n, c * d, h, w), channels-last memory
        return outputs

    def _prefetch_geometry(self, x, wanted, last):
        """The strided geometries of this forward pass, asked for up front (spconv.core.prefetch_downsample_chain): the first
        strided convolution of the stem and of every stage that will run form a chain (submanifold layers keep the sites), the
        z-collapsing heads branch off the levels they read."""
        from ...spconv import core

        def first_strided(module):
            for m in module.modules():
                if isinstance(m, SparseConv3d) and not m.subm and any(int(v) != 1 for v in m.stride):
                    return (m.kernel_size, m.stride, m.padding)
            return None

        chain, level_of = [], {}
        spec = first_strided(self.stem)
        if spec is not None:
            chain.append(spec)
        level_of["stem"] = len(chain)
        for i, (stage, name) in enumerate(self.stages_and_names):
            if i > last:
                break
            spec = first_strided(stage)
            if spec is not None:
                chain.append(spec)
            level_of[name] = len(chain)
        branches = []
        for name in wanted:
            spec = first_strided(getattr(self, name + "_out"))
            if spec is not None and name in level_of:
                branches.append((level_of[name], spec))
        core.prefetch_downsample_chain(x, chain, branches)

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def build_sparse_resnet_backbone(config, in_channels):
    """res18 / res34 (basic blocks) and res50 (bottleneck blocks of num_groups * width_per_group channels, doubling
    per stage) plans of efg/modeling/backbones/sparse_net.py:318-397."""
    depth = _get(config, "depth")
    stem_width = {18: 16, "18b": 24, "18c": 32, 34: 16, "34b": 24, "34c": 32, 50: 16}[depth]
    norm, activation = _get(config, "norm"), _get(config, "activation")
    stem = SparseBasicStem(in_channels=in_channels, out_channels=_get(config, "stem_out_channels"), norm=norm,
                           activation=activation, stem_width=stem_width, indice_key="stem")
    out_features = list(_get(config, "out_features"))
    in_channels = _get(config, "stem_out_channels")
    out_channels = _get(config, "res1_out_channels")
    num_blocks_per_stage = {18: [2, 2, 2, 2], "18b": [2, 2, 2, 2], "18c": [2, 2, 2, 2], 34: [3, 4, 6, 3],
                            "34b": [3, 4, 6, 3], "34c": [3, 4, 6, 3], 50: [3, 4, 6, 3]}[depth]
    bottleneck = depth not in (18, "18b", "18c", 21, 34)   # (:380-387: basic blocks for these, bottleneck blocks otherwise -- "34b" / "34c" too)
    bottleneck_channels = _get(config, "num_groups") * _get(config, "width_per_group") if bottleneck else None
    max_stage_idx = max({"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in out_features)
    stages = []
    for idx, stage_idx in enumerate(range(2, max_stage_idx + 1)):
        if bottleneck:
            blocks = make_stage(SparseBottleneckBlock, num_blocks_per_stage[idx], 2, in_channels=in_channels,
                                out_channels=out_channels, bottleneck_channels=bottleneck_channels, norm=norm,
                                activation=activation, indice_key="res" + str(stage_idx))
            bottleneck_channels *= 2
        else:
            blocks = make_stage(SparseBasicResBlock, num_blocks_per_stage[idx], 2, in_channels=in_channels,
                                out_channels=out_channels, norm=norm, activation=activation,
                                indice_key="res" + str(stage_idx))
        in_channels = out_channels
        out_channels *= 2
        stages.append(blocks)
    return SparseResNet(stem, stages, out_features=out_features, norm=norm)


class SparseBasicBlock(spconv.SparseModule):
    """CenterPoint residual block: 2 x SubMConv3d(k3) sharing `indice_key` (:429-470)."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm="BN1d", downsample=None, indice_key=None):
        super().__init__()
        bias = norm is not None
        self.conv1 = SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                                indice_key=indice_key)
        self.bn1 = common.get_norm(norm, planes)
        self.relu = nn.ReLU()
        self.conv2 = SubMConv3d(planes, planes, kernel_size=3, stride=1, padding=1, bias=bias, indice_key=indice_key)
        self.bn2 = common.get_norm(norm, planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out = out.replace_feature(bn_act(out.features, self.bn1, relu=True))
        out = self.conv2(out)
        if self.downsample is not None:
            identity = self.downsample(x)
        # relu(bn2(.) + identity) in one pass (HIP BatchNorm + residual + ReLU on the GPU in training; the PyTorch
        # modules otherwise)
        return out.replace_feature(bn_act(out.features, self.bn2, relu=True, residual=identity.features))


class SpMiddleResNetFHD(nn.Module):
    """CenterPoint middle encoder (:473-545): SubM stem, 4 levels of 2 residual blocks joined by
    strided SparseConv3d, z-collapsing `extra_conv`, dense [B, 128*D, H, W] output."""

    def __init__(self, num_input_features=128, out_features=("res3",), norm="BN1d"):
        super().__init__()
        self.conv_input = spconv.SparseSequential(
            SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"),
            common.get_norm(norm, 16), nn.ReLU(inplace=True))
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(16, 16, norm=norm, indice_key="res0"),
            SparseBasicBlock(16, 16, norm=norm, indice_key="res0"))
        self.conv2 = spconv.SparseSequential(
            SparseConv3d(16, 32, 3, 2, padding=1, bias=False), common.get_norm(norm, 32), nn.ReLU(inplace=True),
            SparseBasicBlock(32, 32, norm=norm, indice_key="res1"),
            SparseBasicBlock(32, 32, norm=norm, indice_key="res1"))
        self.conv3 = spconv.SparseSequential(
            SparseConv3d(32, 64, 3, 2, padding=1, bias=False), common.get_norm(norm, 64), nn.ReLU(inplace=True),
            SparseBasicBlock(64, 64, norm=norm, indice_key="res2"),
            SparseBasicBlock(64, 64, norm=norm, indice_key="res2"))
        self.conv4 = spconv.SparseSequential(
            SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False), common.get_norm(norm, 128),
            nn.ReLU(inplace=True),
            SparseBasicBlock(128, 128, norm=norm, indice_key="res3"),
            SparseBasicBlock(128, 128, norm=norm, indice_key="res3"))
        self.extra_conv = spconv.SparseSequential(
            SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), common.get_norm(norm, 128), nn.ReLU())

    def forward(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        ret = spconv.SparseConvTensor(voxel_features, coors.int(), sparse_shape.tolist(), batch_size)
        x = self.conv_input(ret)
        x = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        return self.extra_conv(x).dense_bev()  # == .dense().view(n, c * d, h, w), channels-last memory
