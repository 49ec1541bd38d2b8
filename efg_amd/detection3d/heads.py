"""Per-layer detection heads + loss/metric plumbing ($CQ/heads.py:14-96, $CQ/modules/metrics.py)."""
import math

import torch
from torch import nn

from .losses import Det3DLoss
from .matcher import HungarianMatcher3d
from ..operators.det_loss import box_refine
from .utils import MLP, get_clones


def accuracy(output, target):
    """top-1 precision in percent ($CQ/modules/metrics.py:33-54)."""
    if target.numel() == 0:
        return torch.zeros([], device=output.device)
    pred = output.argmax(1)
    return pred.eq(target).float().sum() * (100.0 / target.size(0))


class Det3DHead(nn.Module):
    def __init__(self, config, with_aux=False, with_metrics=False, num_classes=3, num_layers=1):
        super().__init__()
        hidden_dim = config.model.hidden_dim
        class_embed = MLP(hidden_dim, hidden_dim, num_classes, 3)
        bbox_embed = MLP(hidden_dim, hidden_dim, 7, 3)
        prior_prob = 0.01
        class_embed.layers[-1].bias.data = torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))
        nn.init.constant_(bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(bbox_embed.layers[-1].bias.data, 0)
        self.class_embed = get_clones(class_embed, num_layers)
        self.bbox_embed = get_clones(bbox_embed, num_layers)
        mc = config.model.loss.matcher
        matcher = HungarianMatcher3d(cost_class=mc.class_weight, cost_bbox=mc.bbox_weight, cost_giou=mc.giou_weight,
                                     cost_rad=mc.rad_weight)
        weight_dict = {"loss_ce": config.model.loss.class_loss_coef, "loss_bbox": config.model.loss.bbox_loss_coef,
                       "loss_giou": config.model.loss.giou_loss_coef, "loss_rad": config.model.loss.rad_loss_coef}
        self.losses = Det3DLoss(matcher=matcher, weight_dict=weight_dict, losses=["focal_labels", "boxes"])
        if with_aux:
            aux = {k + "_enc_0": v for k, v in self.losses.weight_dict.items()}
            for i in range(config.model.transformer.dec_layers - 1):
                aux.update({k + f"_{i}": v for k, v in self.losses.weight_dict.items()})
            self.losses.weight_dict.update(aux)
        self.with_metrics = with_metrics
        self.config = config

    def forward(self, embed, anchors, layer_idx=0):
        cls_logits = self.class_embed[layer_idx](embed)
        box_coords = box_refine(self.bbox_embed[layer_idx](embed), anchors)   # (delta + inverse_sigmoid(anchors)).sigmoid()
        return cls_logits, box_coords

    def compute_losses(self, outputs, targets, dn_meta=None, prepared=None, q_of_g=None):
        # weighted in the loss module, all terms in one multiply (*_dn / *_dn_i keys are not in the dict -> weight 1, as in
        # the reference's loop over weight_dict); `prepared` / `q_of_g`: the caller matched this call already (losses.match_together)
        loss_dict = self.losses(outputs, targets, dn_meta=dn_meta, weights=self.losses.weight_dict, prepared=prepared,
                                q_of_g=q_of_g)
        if self.with_metrics:
            with torch.no_grad():
                loss_dict["accuracy"] = accuracy(*self.losses.get_target_classes())
        return loss_dict
