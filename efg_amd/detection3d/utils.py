"""Box / misc helpers of the ConQueR playground ($CQ/modules/utils.py:16-113,277-314, $CQ/modules/blocks.py)."""
import copy

import torch
import torch.nn.functional as F
from torch import nn

from ..operators.linear import Linear, linear


class MLP(nn.Module):
    """$CQ/modules/blocks.py:5-17: Linear+ReLU stack, no activation on the last layer."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            # hidden layers: Linear + ReLU as one product with a fused epilogue / one-launch backward (operators/linear.py)
            x = linear(x, layer.weight, layer.bias, relu=True) if i < self.num_layers - 1 else layer(x)
        return x


def get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def inverse_sigmoid(x, eps=1e-5):
    """$CQ/modules/utils.py:83-87."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def box_cxcyczlwh_to_xyxyxy(x):
    c, s = x[..., :3], x[..., 3:6]
    return torch.cat((c - 0.5 * s, c + 0.5 * s), dim=-1)


def generalized_box3d_iou(boxes1, boxes2):
    """Axis-aligned 3-D GIoU matrix [N, M] of (x0,y0,z0,x1,y1,z1) boxes ($CQ/modules/utils.py:29-72)."""
    boxes1 = torch.nan_to_num(boxes1)
    boxes2 = torch.nan_to_num(boxes2)
    vol1 = (boxes1[:, 3] - boxes1[:, 0]) * (boxes1[:, 4] - boxes1[:, 1]) * (boxes1[:, 5] - boxes1[:, 2])
    vol2 = (boxes2[:, 3] - boxes2[:, 0]) * (boxes2[:, 4] - boxes2[:, 1]) * (boxes2[:, 5] - boxes2[:, 2])
    lo = torch.max(boxes1[:, None, :3], boxes2[:, :3])
    hi = torch.min(boxes1[:, None, 3:], boxes2[:, 3:])
    lwh = (hi - lo).clamp(min=0)
    inter = lwh[:, :, 0] * lwh[:, :, 1] * lwh[:, :, 2]
    union = vol1[:, None] + vol2 - inter
    iou = inter / union
    lo = torch.min(boxes1[:, None, :3], boxes2[:, :3])
    hi = torch.max(boxes1[:, None, 3:], boxes2[:, 3:])
    whl = (hi - lo).clamp(min=0)
    vol = whl[:, :, 0] * whl[:, :, 1] * whl[:, :, 2]
    return iou - (vol - union) / vol


def pairwise_box3d_giou(boxes1, boxes2):
    """generalized_box3d_iou with leading batch dims: boxes1 [..., N, 6], boxes2 [..., M, 6] -> [..., N, M]."""
    boxes1 = torch.nan_to_num(boxes1)[..., :, None, :]
    boxes2 = torch.nan_to_num(boxes2)[..., None, :, :]
    vol1 = (boxes1[..., 3] - boxes1[..., 0]) * (boxes1[..., 4] - boxes1[..., 1]) * (boxes1[..., 5] - boxes1[..., 2])
    vol2 = (boxes2[..., 3] - boxes2[..., 0]) * (boxes2[..., 4] - boxes2[..., 1]) * (boxes2[..., 5] - boxes2[..., 2])
    lwh = (torch.min(boxes1[..., 3:], boxes2[..., 3:]) - torch.max(boxes1[..., :3], boxes2[..., :3])).clamp(min=0)
    inter = lwh[..., 0] * lwh[..., 1] * lwh[..., 2]
    union = vol1 + vol2 - inter
    whl = (torch.max(boxes1[..., 3:], boxes2[..., 3:]) - torch.min(boxes1[..., :3], boxes2[..., :3])).clamp(min=0)
    vol = whl[..., 0] * whl[..., 1] * whl[..., 2]
    return inter / union - (vol - union) / vol


def paired_box3d_giou(boxes1, boxes2):
    """Row-wise GIoU (the diagonal of generalized_box3d_iou, without building the matrix)."""
    boxes1 = torch.nan_to_num(boxes1)
    boxes2 = torch.nan_to_num(boxes2)
    vol1 = (boxes1[:, 3] - boxes1[:, 0]) * (boxes1[:, 4] - boxes1[:, 1]) * (boxes1[:, 5] - boxes1[:, 2])
    vol2 = (boxes2[:, 3] - boxes2[:, 0]) * (boxes2[:, 4] - boxes2[:, 1]) * (boxes2[:, 5] - boxes2[:, 2])
    lwh = (torch.min(boxes1[:, 3:], boxes2[:, 3:]) - torch.max(boxes1[:, :3], boxes2[:, :3])).clamp(min=0)
    inter = lwh[:, 0] * lwh[:, 1] * lwh[:, 2]
    union = vol1 + vol2 - inter
    whl = (torch.max(boxes1[:, 3:], boxes2[:, 3:]) - torch.min(boxes1[:, :3], boxes2[:, :3])).clamp(min=0)
    vol = whl[:, 0] * whl[:, 1] * whl[:, 2]
    return inter / union - (vol - union) / vol


def sigmoid_focal_loss(logits, targets, alpha=-1.0, gamma=2.0, reduction="none"):
    """efg/modeling/losses/focal_loss.py:5-45."""
    p = torch.sigmoid(logits)
    ce_loss = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


_SHAPES = {}


def _shapes_on_device(shapes, device):
    """int64 [N,2] level shapes on `device`, built once per distinct shape set: torch.tensor(list, device=...)
    is a SYNCHRONOUS upload -- it waits for everything queued on the stream (the previous step's backward)."""
    key = (shapes, str(device))
    if key not in _SHAPES:
        _SHAPES[key] = torch.tensor(shapes, dtype=torch.int64).to(device)
    return _SHAPES[key]


def flatten_with_shape(tensor_list):
    """[(B,C,Hi,Wi)] -> ((B, sum Hi*Wi, C), int64 [N,2] shapes) ($CQ/modules/utils.py:277-314)."""
    shapes = _shapes_on_device(tuple((t.shape[2], t.shape[3]) for t in tensor_list), tensor_list[0].device)
    if len(tensor_list) == 1:  # one level: a channels-last map already IS the token layout (cat would copy it)
        flat = tensor_list[0].flatten(2).permute(0, 2, 1).contiguous()
    else:
        flat = torch.cat([t.flatten(2).permute(0, 2, 1) for t in tensor_list], dim=1)
    return flat, shapes


def limit_period(val, offset=0.5, period=3.141592653589793):
    """efg/geometry/box_ops_torch.py:229-241."""
    return val - torch.floor(val / period + offset) * period
