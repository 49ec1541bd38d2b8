"""Norm / activation factories and the Conv2d(+norm) wrapper (efg/modeling/common/batch_norm.py:140-188,
efg/modeling/common/blocks.py:45-100).  Dense ops stay on PyTorch-ROCm (MIOpen / hipBLASLt)."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from ...operators import batchnorm as _bn
from ...operators import linear as _linear_mod
from ...operators.conv2d import arm_covers, conv3x3, conv3x3_arm, deterministic_mode
from ...operators.linear import linear


def get_norm(norm, out_channels):
    """efg/modeling/common/batch_norm.py:140-168.  "BN" / "BN1d" / "GN" (32 groups); "" -> None."""
    args = None
    if isinstance(norm, (list, tuple)):
        norm, args = norm
    if isinstance(norm, str):
        if len(norm) == 0:
            return None
        norm = {
            "BN": nn.BatchNorm2d,
            "BN1d": nn.BatchNorm1d,
            "GN": lambda channels: nn.GroupNorm(32, channels),
            "nnSyncBN": nn.SyncBatchNorm,
        }[norm]
    return norm(out_channels, **args) if args else norm(out_channels)


def get_activation(activation):
    """efg/modeling/common/batch_norm.py:171-188: {type: ReLU|ReLU6, inplace: bool} or None."""
    if activation is None:
        return None
    atype = activation["type"] if isinstance(activation, dict) else activation.type
    inplace = activation["inplace"] if isinstance(activation, dict) else activation.inplace
    return {"ReLU": nn.ReLU, "ReLU6": nn.ReLU6}[atype](inplace=inplace)


class Conv2d(nn.Conv2d):
    """nn.Conv2d followed by an optional norm and activation (efg/modeling/common/blocks.py:45-100)."""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def _is_pointwise(self):
        return (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
                and self.dilation == (1, 1) and self.groups == 1)

    def forward(self, x):
        if self._is_pointwise():
            # 1x1 conv == Linear over the channel axis of the channels-last map: goes to hipBLASLt
            # (MIOpen's 1x1 weight-gradient picks a single-workgroup GEMM for K = B*H*W ~ 70k rows).
            y = linear(x.permute(0, 2, 3, 1), self.weight.view(self.out_channels, self.in_channels), self.bias)
            x = y.permute(0, 3, 1, 2)
        elif (x.is_cuda and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
              and self.dilation == (1, 1) and self.groups == 1 and torch.is_grad_enabled() and deterministic_mode()):
            x = conv3x3(x, self.weight, self.bias)   # EFG_DETERMINISTIC=1: the layer off MIOpen, as fixed-order GEMMs
        elif (x.is_cuda and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
              and self.dilation == (1, 1) and self.groups == 1 and torch.is_grad_enabled() and _linear_mod._ARM_BF16X3
              and os.environ.get("EFG_CONV2D_ARM", "1") != "0" and arm_covers(x, self.weight)):
            x = conv3x3_arm(x, self.weight, self.bias)   # the split-precision A/B arm (never the default)
        else:
            x = super().forward(x)
        if self.norm is not None:
            if _bn.fusable_nhwc(self.norm, x):
                # a training-mode BatchNorm2d over a channels-last map (+ ReLU): the [B*H*W, C] row kernels of the sparse
                # backbone (operators/batchnorm.py) instead of MIOpen's batch norm + a separate ReLU each way
                relu = type(self.activation) is nn.ReLU
                x = _bn.bn_act_nhwc(x, self.norm, relu)
                if relu:
                    return x
            else:
                x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def c2_xavier_fill(module):
    """efg/modeling/common/weight_init.py:52-64 (Caffe2 XavierFill == kaiming_uniform_(a=1))."""
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)
