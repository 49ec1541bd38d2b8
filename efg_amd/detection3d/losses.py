"""Set-prediction losses of Voxel-DETR / ConQueR ($CQ/losses.py): focal classification, L1 box /
angle, axis-aligned 3-D GIoU, for matched queries, auxiliary layers and denoising groups."""
import torch
from torch import nn
from torch.nn import functional as F

from .utils import box_cxcyczlwh_to_xyxyxy, paired_box3d_giou, sigmoid_focal_loss


def get_world_size():
    return torch.distributed.get_world_size() if (torch.distributed.is_available()
                                                  and torch.distributed.is_initialized()) else 1


def _src_permutation_idx(indices):
    batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
    src_idx = torch.cat([src for (src, _) in indices])
    return batch_idx, src_idx


class ClassificationLoss(nn.Module):
    """$CQ/losses.py:26-73."""

    def __init__(self, focal_alpha):
        super().__init__()
        self.focal_alpha = focal_alpha
        self.target_classes = None
        self.src_logits = None

    def forward(self, outputs, targets, indices, num_boxes):
        outputs["matched_indices"] = indices
        src_logits = outputs["pred_logits"]
        dev = src_logits.device
        target_onehot = torch.zeros_like(src_logits)
        idx = tuple(t.to(dev) for t in _src_permutation_idx(indices))
        target_classes_o = torch.cat([t["labels"][J.to(dev)] for t, (_, J) in zip(targets, indices)])
        self.target_classes = target_classes_o
        if "topk_indexes" in outputs:
            topk = outputs["topk_indexes"]
            self.src_logits = torch.gather(src_logits, 1, topk.expand(-1, -1, src_logits.shape[-1]))[idx]
            target_onehot[idx[0], topk[idx].squeeze(-1), target_classes_o] = 1
        else:
            self.src_logits = src_logits[idx]
            target_onehot[idx[0], idx[1], target_classes_o] = 1
        loss_ce = sigmoid_focal_loss(src_logits, target_onehot, alpha=self.focal_alpha, gamma=2.0,
                                     reduction="sum") / num_boxes
        return {"loss_ce": loss_ce}


class RegressionLoss(nn.Module):
    """$CQ/losses.py:76-108."""

    def forward(self, outputs, targets, indices, num_boxes):
        dev = outputs["pred_boxes"].device
        idx = tuple(t.to(dev) for t in _src_permutation_idx(indices))
        if "topk_indexes" in outputs:
            pred_boxes = torch.gather(outputs["pred_boxes"], 1,
                                      outputs["topk_indexes"].expand(-1, -1, outputs["pred_boxes"].shape[-1]))
        else:
            pred_boxes = outputs["pred_boxes"]
        target_boxes = torch.cat([t["gt_boxes"][i.to(dev)] for t, (_, i) in zip(targets, indices)], dim=0)
        src_boxes, src_rads = pred_boxes[idx].split(6, dim=-1)
        target_boxes, target_rads = target_boxes.split(6, dim=-1)
        loss_bbox = F.l1_loss(src_boxes, target_boxes, reduction="none")
        loss_rad = F.l1_loss(src_rads, target_rads, reduction="none")
        # 1 - diag(GIoU matrix) of the reference (:95-100), computed pairwise
        loss_giou = 1 - paired_box3d_giou(box_cxcyczlwh_to_xyxyxy(src_boxes), box_cxcyczlwh_to_xyxyxy(target_boxes))
        return {"loss_bbox": loss_bbox.sum() / num_boxes, "loss_giou": loss_giou.sum() / num_boxes,
                "loss_rad": loss_rad.sum() / num_boxes}


class Det3DLoss(nn.Module):
    """$CQ/losses.py:111-214."""

    def __init__(self, matcher, weight_dict, losses):
        super().__init__()
        self.matcher, self.weight_dict, self.losses = matcher, weight_dict, losses
        self.det3d_losses = nn.ModuleDict()
        self.det3d_enc_losses = nn.ModuleDict()
        for loss in losses:
            if loss == "boxes":
                self.det3d_losses[loss] = RegressionLoss()
                self.det3d_enc_losses[loss + "_enc"] = RegressionLoss()
            elif loss == "focal_labels":
                self.det3d_losses[loss] = ClassificationLoss(0.25)
                self.det3d_enc_losses[loss + "_enc"] = ClassificationLoss(0.25)
            else:
                raise ValueError(f"Only boxes|focal_labels are supported for det3d losses. Found {loss}")

    def get_target_classes(self):
        for k in self.det3d_losses.keys():
            if "labels" in k:
                return self.det3d_losses[k].src_logits, self.det3d_losses[k].target_classes

    def forward(self, outputs, targets, dn_meta=None):
        dev = next(iter(outputs.values())).device
        num_boxes = torch.as_tensor([sum(len(t["labels"]) for t in targets)], dtype=torch.float, device=dev)
        if get_world_size() > 1:
            torch.distributed.all_reduce(num_boxes)
        num_boxes = torch.clamp(num_boxes / get_world_size(), min=1).item() if get_world_size() > 1 else max(
            float(sum(len(t["labels"]) for t in targets)), 1.0)
        losses = {}
        if dn_meta is not None:
            known = dn_meta["output_known_lbs_bboxes"]
            scalar, pad_size = dn_meta["num_dn_group"], dn_meta["pad_size"]
            assert pad_size % scalar == 0
            single_pad = pad_size // scalar
            dn_pos_idx = []
            for tgt in targets:
                n = len(tgt["labels"])
                if n > 0:
                    # quirk kept: arange(0, n - 1) leaves the LAST GT of every scene out of the DN loss (:161)
                    t = torch.arange(0, n - 1).long().unsqueeze(0).repeat(scalar, 1)
                    tgt_idx = t.flatten()
                    output_idx = ((torch.arange(scalar) * single_pad).long().unsqueeze(1) + t).flatten()
                else:
                    output_idx = tgt_idx = torch.tensor([]).long()
                dn_pos_idx.append((output_idx, tgt_idx))
            l_dict = {}
            for loss in self.losses:
                l_dict.update(self.det3d_losses[loss](known, targets, dn_pos_idx, num_boxes * scalar))
            losses.update({k + "_dn": v for k, v in l_dict.items()})
        if "aux_outputs" in outputs:
            for i, aux_outputs in enumerate(outputs["aux_outputs"]):
                indices = self.matcher(aux_outputs, targets)
                for loss in self.losses:
                    l_dict = self.det3d_losses[loss](aux_outputs, targets, indices, num_boxes)
                    losses.update({k + f"_{i}": v for k, v in l_dict.items()})
                if dn_meta is not None:
                    aux_known = known["aux_outputs"][i]
                    l_dict = {}
                    for loss in self.losses:
                        l_dict.update(self.det3d_losses[loss](aux_known, targets, dn_pos_idx, num_boxes * scalar))
                    losses.update({k + f"_dn_{i}": v for k, v in l_dict.items()})
        indices = self.matcher(outputs, targets)
        for loss in self.losses:
            losses.update(self.det3d_losses[loss](outputs, targets, indices, num_boxes))
        return losses
