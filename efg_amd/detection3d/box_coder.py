"""GT box normalisation to [0,1] codes and back ($CQ/modules/box_coder.py:35-80)."""
import numpy as np
import torch

from .utils import limit_period


class VoxelBoxCoder3D:
    def __init__(self, voxel_size, pc_range, n_dim=7, device=torch.device("cpu")):
        self.device = device
        self.voxel_size = torch.tensor(voxel_size, device=device)
        self.pc_range = torch.tensor(pc_range, device=device)
        self.pc_size = self.pc_range[3:] - self.pc_range[:3]
        self.z_normalizer = 10.0
        self.n_dim = n_dim

    @property
    def code_size(self):
        return self.n_dim

    def encode(self, target):
        """In-place on the target dict like the reference: labels 1..3 -> 0..2; boxes
        (x,y,z,l,w,h,...,yaw) -> 7 codes in [0,1]."""
        target["labels"] -= 1
        b = target["gt_boxes"]
        b[:, :2] -= self.pc_range[:2]
        b[:, :2] /= self.pc_size[:2]
        b[:, 2] -= -1 * self.z_normalizer
        b[:, 2] /= 2 * self.z_normalizer
        b[:, 3:5] /= self.pc_size[:2]
        b[:, 5] /= 2 * self.z_normalizer
        b[:, -1] = limit_period(b[:, -1], offset=0.5, period=np.pi * 2)
        b = b[:, [0, 1, 2, 3, 4, 5, -1]]
        b[:, -1] = (b[:, -1] + 0.5 * np.pi * 2) / (np.pi * 2)  # normalize_period, $CQ/modules/utils.py:79-80
        target["gt_boxes"] = b
        assert ((b >= 0) & (b <= 1)).all().item()
        return target

    def decode(self, pred_boxes):
        pred_boxes[..., :2] = pred_boxes[..., :2] * self.pc_size[:2] + self.pc_range[:2]
        pred_boxes[..., 2] = pred_boxes[..., 2] * 2 * self.z_normalizer + -1 * self.z_normalizer
        pred_boxes[..., 3:5] = pred_boxes[..., 3:5] * self.pc_size[:2]
        pred_boxes[..., 5] = pred_boxes[..., 5] * 2 * self.z_normalizer
        pred_boxes[..., -1] = pred_boxes[..., -1] * np.pi * 2 - np.pi
        return pred_boxes
