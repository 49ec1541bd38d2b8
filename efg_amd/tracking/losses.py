"""Regression losses of TrajectoryFormer (counterpart of $TF/losses.py)."""
import numpy as np
import torch
from torch import nn

from .geometry import boxes_to_corners_3d, decode_torch, rotate_points_along_z


class WeightedSmoothL1Loss(nn.Module):
    """Element-wise smooth-L1 with optional per-code and per-anchor weights, unreduced; NaN targets are ignored
    (losses.py:7-76)."""

    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta
        self.code_weights = None if code_weights is None else torch.as_tensor(np.asarray(code_weights, np.float32))

    @staticmethod
    def smooth_l1_loss(diff, beta):
        n = torch.abs(diff)
        if beta < 1e-5:
            return n
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)
        diff = input - target
        if self.code_weights is not None:
            diff = diff * self.code_weights.to(diff.device).view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


def get_corner_loss_lidar(pred_bbox3d, gt_bbox3d):
    """(N, 7) x (N, 7) -> (N,): smooth-L1 of the corner distances, the smaller of the box and its 180-degree flip
    (losses.py:79-103)."""
    assert pred_bbox3d.shape[0] == gt_bbox3d.shape[0]
    pred = boxes_to_corners_3d(pred_bbox3d)
    flipped = torch.cat([gt_bbox3d[:, :6], gt_bbox3d[:, 6:7] + np.pi], dim=1)
    dist = torch.min(torch.norm(pred - boxes_to_corners_3d(gt_bbox3d), dim=2),
                     torch.norm(pred - boxes_to_corners_3d(flipped), dim=2))
    return WeightedSmoothL1Loss.smooth_l1_loss(dist, beta=1.0).mean(dim=1)


def get_corner_loss(rcnn_reg, roi_boxes3d, gt_of_rois_src, fg):
    """Decode the foreground residuals in their ROI frames, move them back to the scene, corner loss against the
    ground truth (losses.py:106-130).  `fg`: the reference's boolean mask, or the (ascending) row indices it selects
    -- no `nonzero` read-back per selection then.  Mean over an empty foreground set is NaN, as in the reference."""
    if fg.dtype == torch.bool:
        fg = fg.nonzero().view(-1)
    rois = roi_boxes3d.index_select(0, fg)
    anchors = torch.cat([torch.zeros_like(rois[:, :3]), rois[:, 3:]], dim=1).detach()
    boxes = decode_torch(rcnn_reg.index_select(0, fg), anchors)
    boxes = rotate_points_along_z(boxes.unsqueeze(1), rois[:, 6]).squeeze(1)
    boxes = torch.cat([boxes[:, 0:3] + rois[:, 0:3], boxes[:, 3:]], dim=1)
    return get_corner_loss_lidar(boxes[:, 0:7], gt_of_rois_src.index_select(0, fg)[:, 0:7]).mean()
