"""Hungarian matching of queries to GT boxes ($CQ/modules/matcher.py:9-96).  The cost matrices are built on
the GPU for all layers and scenes at once and -- in the training step (`match_layers`) -- assigned there too
(efg_lsap_f32 reproduces scipy's linear_sum_assignment exactly); the reference moves one matrix per scene to the
host and calls scipy.  `forward` keeps the reference's call format (one transfer, scipy)."""
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from ..operators.assignment import linear_sum_assignment_batched
from ..operators.det_loss import match_cost
from .utils import box_cxcyczlwh_to_xyxyxy, generalized_box3d_iou, pairwise_box3d_giou


class HungarianMatcher3d(nn.Module):
    def __init__(self, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0, cost_rad=1.0):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou, self.cost_rad = cost_class, cost_bbox, cost_giou, cost_rad
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0 or cost_rad != 0, "all costs cant be 0"

    @torch.no_grad()
    def cost_matrices(self, outputs, targets):
        if "topk_indexes" in outputs:
            idx = outputs["topk_indexes"]
            pred_logits = torch.gather(outputs["pred_logits"], 1, idx.expand(-1, -1, outputs["pred_logits"].shape[-1]))
            pred_boxes = torch.gather(outputs["pred_boxes"], 1, idx.expand(-1, -1, outputs["pred_boxes"].shape[-1]))
        else:
            pred_logits, pred_boxes = outputs["pred_logits"], outputs["pred_boxes"]
        bs = pred_logits.shape[0]
        out_prob = pred_logits.sigmoid().float()
        out_bbox, out_rad = pred_boxes.float().split(6, dim=-1)
        alpha, gamma = 0.25, 2.0
        neg_cost = (1 - alpha) * (out_prob ** gamma) * (-(1 - out_prob + 1e-8).log())
        pos_cost = alpha * ((1 - out_prob) ** gamma) * (-(out_prob + 1e-8).log())
        mats = []
        for i in range(bs):
            tgt_ids = targets[i]["labels"]
            tgt_bbox = targets[i]["gt_boxes"][..., :6].float()
            tgt_rad = targets[i]["gt_boxes"][..., 6:].float()
            cost_giou = -generalized_box3d_iou(box_cxcyczlwh_to_xyxyxy(out_bbox[i]), box_cxcyczlwh_to_xyxyxy(tgt_bbox))
            cost_class = pos_cost[i][:, tgt_ids] - neg_cost[i][:, tgt_ids]
            cost_bbox = torch.cdist(out_bbox[i], tgt_bbox, p=1)
            cost_rad = torch.cdist(out_rad[i], tgt_rad, p=1)
            mats.append(self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou +
                        self.cost_rad * cost_rad)
        return mats

    @torch.no_grad()
    def match_layers(self, logits, boxes, tgt_labels, tgt_boxes, counts):
        """All decoder layers and scenes at once.  logits [L,B,Q,C], boxes [L,B,Q,7]; padded targets
        tgt_labels [B,G], tgt_boxes [B,G,7] with `counts[b]` valid columns.  Same cost as `cost_matrices`
        ($CQ/modules/matcher.py:40-80).  Returns query_of_gt int64 [L,B,G] on the inputs' device: the query
        matched to GT column g (-1 in padded columns) -- the reference's (row_ind, col_ind) pairs of
        (layer l, scene b) are {(query_of_gt[l,b,g], g)}.  On the GPU the assignment is efg_lsap_f32 (no
        host transfer); on CPU tensors it is scipy, as in the reference."""
        n_layers, bs, nq = logits.shape[:3]
        g = tgt_labels.shape[1]
        if logits.is_cuda:  # cost in one kernel (csrc/det_loss.hip), assignment on the device (csrc/matcher.hip)
            cost = match_cost(logits, boxes, tgt_labels, tgt_boxes, self.cost_class, self.cost_bbox, self.cost_giou,
                              self.cost_rad)
            ng = torch.tensor(list(counts) * n_layers, dtype=torch.int32).to(cost.device, non_blocking=True)
            return linear_sum_assignment_batched(cost, ng).view(n_layers, bs, g)
        out_prob = logits.sigmoid().float()
        out_bbox, out_rad = boxes.float().split(6, dim=-1)
        alpha, gamma = 0.25, 2.0
        neg_cost = (1 - alpha) * (out_prob ** gamma) * (-(1 - out_prob + 1e-8).log())
        pos_cost = alpha * ((1 - out_prob) ** gamma) * (-(out_prob + 1e-8).log())
        lab = tgt_labels[None, :, None, :].expand(n_layers, bs, nq, g)
        cost_class = torch.gather(pos_cost, 3, lab) - torch.gather(neg_cost, 3, lab)
        tb, tr = tgt_boxes[..., :6].float(), tgt_boxes[..., 6:].float()
        cost_bbox = (out_bbox[:, :, :, None, :] - tb[None, :, None, :, :]).abs().sum(-1)
        cost_rad = (out_rad[:, :, :, None, :] - tr[None, :, None, :, :]).abs().sum(-1)
        cost_giou = -pairwise_box3d_giou(box_cxcyczlwh_to_xyxyxy(out_bbox), box_cxcyczlwh_to_xyxyxy(tb)[None])
        cost = (self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou +
                self.cost_rad * cost_rad)
        host = cost.numpy()
        q_of_g = torch.full((n_layers, bs, g), -1, dtype=torch.int64)
        for li in range(n_layers):
            for b in range(bs):
                i, j = linear_sum_assignment(host[li, b, :, : counts[b]])
                q_of_g[li, b, torch.as_tensor(j, dtype=torch.int64)] = torch.as_tensor(i, dtype=torch.int64)
        return q_of_g

    @torch.no_grad()
    def forward(self, outputs, targets):
        mats = self.cost_matrices(outputs, targets)
        nq = mats[0].shape[0] if mats else 0
        sizes = [m.shape[1] for m in mats]
        host = torch.cat(mats, dim=1).cpu() if mats else None  # the one D2H of this call
        out = []
        for c in (host.split(sizes, dim=1) if mats else []):
            i, j = linear_sum_assignment(c.reshape(nq, -1))
            out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
        return out
