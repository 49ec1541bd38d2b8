"""Set-prediction losses of Voxel-DETR / ConQueR ($CQ/losses.py:26-214): focal classification, L1 box /
angle and axis-aligned 3-D GIoU for matched queries, for every decoder layer (auxiliary outputs) and for
the denoising groups.

Same terms, keys and values as the reference, evaluated in BATCHED form: the reference loops over decoder
layers x {matching, denoising} x {labels, boxes} x scenes with ~40 tiny kernels and one host round trip
each (~1000 launches and 4 syncs per step at 3 layers); here all layers are stacked, matched with ONE
cost-matrix transfer, and each loss family is one pass whose per-layer sums are read off a vector.
"""
import contextlib
import os

import torch
from torch import nn
from torch.nn import functional as F

from ..operators.det_loss import BoxLossLayers, FocalLossLayers, device_scalar
from .utils import box_cxcyczlwh_to_xyxyxy, paired_box3d_giou, sigmoid_focal_loss


def get_world_size():
    return torch.distributed.get_world_size() if (torch.distributed.is_available()
                                                  and torch.distributed.is_initialized()) else 1


class LossDict(dict):
    """The model's loss dictionary (same keys and values as the reference's: one scalar tensor per term) plus the
    VECTORS the scalars are views of.  `total()` sums the vectors: a backward that starts from it never touches the
    ~32 select nodes of the scalar entries (each a zero-fill + copy launch pair in backward), which a
    `sum(loss_dict.values())` as in efg/engine/trainer.py:296-297 would.  Both give the same value up to fp32
    summation order; a trainer that does not know `total()` keeps working."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.vectors = []          # 1-D tensors covering every differentiable entry exactly once

    def add_vector(self, keys, vec):
        """Register `vec` [len(keys)] and expose vec[i] under keys[i]."""
        self.vectors.append(vec)
        for i, k in enumerate(keys):
            self[k] = vec[i]

    def merge(self, other, suffix=""):
        for k, v in other.items():
            self[k + suffix] = v
        self.vectors.extend(getattr(other, "vectors", []))
        return self

    def total(self):
        vs = [v for v in self.vectors if v.requires_grad]
        covered = sum(v.numel() for v in vs)
        n_diff = sum(1 for v in self.values() if torch.is_tensor(v) and v.requires_grad)
        if not vs or covered != n_diff:   # entries added outside add_vector: fall back to the generic sum
            return torch.stack([v for v in self.values() if torch.is_tensor(v) and v.requires_grad]).sum()
        return (vs[0] if len(vs) == 1 else torch.cat(vs)).sum()


class PaddedTargets(list):
    """The per-scene target dicts (a plain list, as the reference passes them) that also carries the padded batch
    form built once on the host: labels [B,G] (0 padded), boxes [B,G,7], counts (host list).  Every dict's
    tensors are views of the padded ones."""

    def __init__(self, labels, boxes, counts):
        super().__init__({"labels": labels[b, :n], "gt_boxes": boxes[b, :n]} for b, n in enumerate(counts))
        self.labels, self.boxes, self.counts = labels, boxes, list(counts)

    def class_agnostic(self):
        """Same boxes, every label 0 (the encoder proposal loss, voxel_detr.py:200-202)."""
        return PaddedTargets(torch.zeros_like(self.labels), self.boxes, self.counts)


def _pad_targets(targets, device):
    """labels [B,G] (0 padded), boxes [B,G,7], counts (host list)."""
    if isinstance(targets, PaddedTargets):
        return targets.labels, targets.boxes, targets.counts
    counts = [int(t["labels"].numel()) for t in targets]
    g = max(max(counts), 1)
    labels = torch.zeros(len(targets), g, dtype=torch.int64, device=device)
    boxes = torch.zeros(len(targets), g, 7, dtype=torch.float32, device=device)
    for b, t in enumerate(targets):
        labels[b, : counts[b]] = t["labels"]
        boxes[b, : counts[b]] = t["gt_boxes"]
    return labels, boxes, counts


def match_together(calls, side=None):
    """calls: [(matcher, prepare() dict), ...] with equal query and (padded) GT counts -> the assignments [L_i, B, G] of every
    call from ONE device-side assignment launch: the Hungarian kernel is one workgroup per (layer, scene) problem and as long
    as its slowest problem, so the encoder-proposal matching (B problems) and the decoder's (L x B) cost one launch's time
    instead of two (2 x ~190 us on the step's critical path).  `side`: a stream to queue the cost and assignment kernels on
    (after everything queued on the current one; the caller joins it) -- the small host-to-device copy of the GT counts stays
    on the CURRENT stream.  None when the calls cannot share a launch (host tensors, different shapes): the caller then lets
    every loss match for itself."""
    from ..operators.assignment import linear_sum_assignment_batched
    from ..operators.det_loss import match_cost

    if not calls or not all(pr["m_logits"].is_cuda for _, pr in calls):
        return None
    q, g = calls[0][1]["m_logits"].shape[2], calls[0][1]["tgt_labels"].shape[1]
    if any(pr["m_logits"].shape[2] != q or pr["tgt_labels"].shape[1] != g for _, pr in calls):
        return None
    dev = calls[0][1]["m_logits"].device
    sizes = [pr["m_logits"].shape[0] * pr["m_logits"].shape[1] for _, pr in calls]
    ng = []
    for _, pr in calls:
        ng += list(pr["counts"]) * pr["m_logits"].shape[0]
    ng_dev = torch.tensor(ng, dtype=torch.int32).to(dev, non_blocking=True)
    main = None
    if side is not None:
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        cost = torch.empty((sum(sizes), q, g), dtype=torch.float32, device=dev)
        off = 0
        for (matcher, pr), n in zip(calls, sizes):
            match_cost(pr["m_logits"], pr["m_boxes"], pr["tgt_labels"], pr["tgt_boxes"], matcher.cost_class, matcher.cost_bbox,
                       matcher.cost_giou, matcher.cost_rad, out=cost[off:off + n])
            off += n
        assigned = linear_sum_assignment_batched(cost, ng_dev)
    if side is not None:
        ng_dev.record_stream(side)
        cost.record_stream(side)
    out, off = [], 0
    for (_, pr), n in zip(calls, sizes):
        out.append(assigned[off:off + n].view(pr["m_logits"].shape[0], pr["m_logits"].shape[1], g))
        off += n
    return out


class Det3DLoss(nn.Module):
    """$CQ/losses.py:111-214 (+ ClassificationLoss :26-73, RegressionLoss :76-108)."""

    def __init__(self, matcher, weight_dict, losses, focal_alpha=0.25):
        super().__init__()
        self.matcher, self.weight_dict, self.losses = matcher, weight_dict, losses
        for loss in losses:
            if loss not in ("boxes", "focal_labels"):
                raise ValueError(f"Only boxes|focal_labels are supported for det3d losses. Found {loss}")
        self.focal_alpha = focal_alpha
        self._metric_logits = self._metric_classes = None

    def get_target_classes(self):
        """(matched logits, target classes) of the LAST matched layer, for the accuracy metric."""
        return self._metric_logits, self._metric_classes

    # ---- one loss family over stacked layers ------------------------------------------------------------
    def _layer_losses(self, logits, boxes, sel, tgt_labels, tgt_boxes, num_boxes, parts=("ce", "box")):
        """logits [L,B,Q,C], boxes [L,B,Q,7]; sel = (l, b, q, g) int64 index vectors of the matched pairs.
        Returns {"loss_ce","loss_bbox","loss_giou","loss_rad"} -> [L] vectors (`parts` selects the families;
        the tensor of an unselected family may be None)."""
        l_idx, b_idx, q_idx, g_idx = sel
        n_layers = (logits if logits is not None else boxes).shape[0]
        out = {}
        cls = tgt_labels[b_idx, g_idx]
        want_ce = "focal_labels" in self.losses and "ce" in parts
        want_box = "boxes" in self.losses and "box" in parts
        if (logits if logits is not None else boxes).is_cuda and os.environ.get("EFG_FUSED_LOSS", "1") != "0":
            # one kernel per loss family and direction (csrc/det_loss.hip) instead of ~40 elementwise launches each
            denom = device_scalar(num_boxes, cls.device)
            if want_ce:
                tcls = torch.full(logits.shape[:-1], -1, dtype=torch.int32, device=logits.device)
                tcls.index_put_((l_idx, b_idx, q_idx), cls.to(torch.int32))
                out["loss_ce"] = FocalLossLayers.apply(logits, tcls, denom, self.focal_alpha, 2.0)
            if want_box:
                sums = BoxLossLayers.apply(boxes, tgt_boxes, l_idx, b_idx, q_idx, g_idx, denom)   # [L, 3]
                # the three families as ONE family-major [3 L] vector under a joint key: three column selects would put
                # three SelectBackward nodes (zero-fill + copy + add each) into the graph, three times per step
                out["loss_bbox|loss_giou|loss_rad"] = sums.t().reshape(-1)
            return out, cls
        if want_ce:
            onehot = torch.zeros_like(logits)
            # (a Python scalar on the right-hand side would be uploaded synchronously)
            onehot.index_put_((l_idx, b_idx, q_idx, cls), logits.new_ones(()))
            fl = sigmoid_focal_loss(logits, onehot, alpha=self.focal_alpha, gamma=2.0, reduction="none")
            out["loss_ce"] = fl.sum(dim=(1, 2, 3)) / num_boxes
        if want_box:
            src = boxes[l_idx, b_idx, q_idx]
            tgt = tgt_boxes[b_idx, g_idx]
            l1 = F.l1_loss(src, tgt, reduction="none")
            giou = 1 - paired_box3d_giou(box_cxcyczlwh_to_xyxyxy(src[:, :6]), box_cxcyczlwh_to_xyxyxy(tgt[:, :6]))
            per = torch.stack((l1[:, :6].sum(1), giou, l1[:, 6:].sum(1)), dim=1)  # [n, 3]
            sums = per.new_zeros(n_layers, 3).index_add_(0, l_idx, per) / num_boxes
            out["loss_bbox"], out["loss_giou"], out["loss_rad"] = sums[:, 0], sums[:, 1], sums[:, 2]
        return out, cls

    def prepare(self, outputs, targets):
        """The layer-stacked tensors the matcher and the losses of `forward` read (auxiliary layers first, the final layer
        last), so that a caller can match SEVERAL loss calls in one assignment launch (`match_together`) and hand each its
        share back through `forward(..., prepared=, q_of_g=)`."""
        dev = outputs["pred_logits"].device
        tgt_labels, tgt_boxes, counts = _pad_targets(targets, dev)
        layers = list(outputs.get("aux_outputs", [])) + [outputs]
        logits = torch.stack([o["pred_logits"] for o in layers])
        topk = outputs.get("topk_indexes")
        if topk is not None:  # encoder proposals: the matcher and the box loss see the gathered top-k set
            assert len(layers) == 1
            m_logits = torch.gather(logits[0], 1, topk.expand(-1, -1, logits.shape[-1]))[None]
            if outputs.get("topk_boxes") is not None:  # boxes evaluated on the top-k tokens only (transformer.py)
                boxes, m_boxes = None, outputs["topk_boxes"][None]
            else:
                boxes = outputs["pred_boxes"][None]
                m_boxes = torch.gather(boxes[0], 1, topk.expand(-1, -1, boxes.shape[-1]))[None]
        else:
            boxes = torch.stack([o["pred_boxes"] for o in layers])
            m_logits, m_boxes = logits, boxes
        return {"layers": layers, "logits": logits, "boxes": boxes, "m_logits": m_logits, "m_boxes": m_boxes, "topk": topk,
                "tgt_labels": tgt_labels, "tgt_boxes": tgt_boxes, "counts": counts}

    def forward(self, outputs, targets, dn_meta=None, weights=None, prepared=None, q_of_g=None):
        """`weights`: {key: coefficient} applied to the terms in one multiply (None: unweighted terms).  `prepared` /
        `q_of_g`: this call's `prepare()` and its assignment [L, B, G], when the caller has matched it already."""
        dev = outputs["pred_logits"].device
        n_gt = sum(len(t["labels"]) for t in targets)
        if get_world_size() > 1:
            # kept on the device (0-dim tensor): the reference reads it back with .item() (losses.py:131-135),
            # which would drain the stream once per step for a value that is only ever a divisor
            nb = torch.tensor([float(n_gt)]).to(dev, non_blocking=True)
            torch.distributed.all_reduce(nb)
            num_boxes = torch.clamp(nb / get_world_size(), min=1)[0]
        else:
            num_boxes = max(float(n_gt), 1.0)
        pr = prepared if prepared is not None else self.prepare(outputs, targets)
        layers, logits, boxes, m_logits, m_boxes, topk = (pr[k] for k in ("layers", "logits", "boxes", "m_logits", "m_boxes", "topk"))
        tgt_labels, tgt_boxes, counts = pr["tgt_labels"], pr["tgt_boxes"], pr["counts"]
        if q_of_g is None:
            q_of_g = self.matcher.match_layers(m_logits, m_boxes, tgt_labels, tgt_boxes, counts)  # [L,B,G] on dev
        outputs["matched_query_of_gt"] = q_of_g[-1]
        # (layer, scene, gt) of every matched pair is known from the GT counts alone: built on the host, one
        # asynchronous upload; the matched query comes from the device-side assignment
        n_layers = len(layers)
        b_l = torch.cat([torch.full((n,), b, dtype=torch.int64) for b, n in enumerate(counts)])
        g_l = torch.cat([torch.arange(n, dtype=torch.int64) for n in counts])
        static = torch.stack([torch.arange(n_layers).repeat_interleave(n_gt), b_l.repeat(n_layers),
                              g_l.repeat(n_layers)]).to(dev, non_blocking=True)
        sel = (static[0], static[1], q_of_g[static[0], static[1], static[2]], static[2])
        if topk is not None:
            # classification targets live on the FULL token set at the positions the top-k picked
            q_full = topk[sel[1], sel[2], 0]
            ce, cls = self._layer_losses(logits, None, (sel[0], sel[1], q_full, sel[3]), tgt_labels, tgt_boxes,
                                         num_boxes, parts=("ce",))
            bx, _ = self._layer_losses(None, m_boxes, sel, tgt_labels, tgt_boxes, num_boxes, parts=("box",))
            per_layer = {"loss_ce": ce["loss_ce"], **{k: v for k, v in bx.items() if k != "loss_ce"}}
            self.__dict__["_metric_logits"] = m_logits[sel[0], sel[1], sel[2]]
        else:
            per_layer, cls = self._layer_losses(logits, boxes, sel, tgt_labels, tgt_boxes, num_boxes)
            # pairs are ordered layer-major: the last layer's are the final n_gt entries
            self.__dict__["_metric_logits"] = logits[-1][sel[1][-n_gt:], sel[2][-n_gt:]] if n_gt else logits[-1][:0, 0]
            cls = cls[-n_gt:] if n_gt else cls[:0]
        self.__dict__["_metric_classes"] = cls   # (plain attributes, set past nn.Module.__setattr__)

        # every family is an [L] vector (aux layers first, the final layer last): the scalar entries are views of it
        fam_keys, fam_vecs = [], []
        n_aux = len(layers) - 1
        for k, v in per_layer.items():
            fam_keys.append([[nm + f"_{i}" for i in range(n_aux)] + [nm] for nm in k.split("|")])   # (joint key: family-major)
            fam_vecs.append(v)

        if dn_meta is not None:
            known = dn_meta["output_known_lbs_bboxes"]
            scalar, pad_size = dn_meta["num_dn_group"], dn_meta["pad_size"]
            assert pad_size % scalar == 0
            single_pad = pad_size // scalar
            dn_layers = list(known.get("aux_outputs", [])) + [known]
            dn_logits = torch.stack([o["pred_logits"] for o in dn_layers])
            dn_boxes = torch.stack([o["pred_boxes"] for o in dn_layers])
            b_l, q_l, g_l = [], [], []
            for b, n in enumerate(counts):
                if n > 0:
                    # quirk kept: arange(0, n - 1) leaves the LAST GT of every scene out of the DN loss (:161)
                    t = torch.arange(0, n - 1).long().unsqueeze(0).repeat(scalar, 1)
                    q_l.append(((torch.arange(scalar) * single_pad).long().unsqueeze(1) + t).flatten())
                    g_l.append(t.flatten())
                    b_l.append(torch.full((t.numel(),), b, dtype=torch.int64))
            if b_l:
                b_i, q_i, g_i = torch.cat(b_l), torch.cat(q_l), torch.cat(g_l)
            else:
                b_i = q_i = g_i = torch.zeros(0, dtype=torch.int64)
            nl = len(dn_layers)
            sel_dn = torch.stack([torch.arange(nl).repeat_interleave(b_i.numel()), b_i.repeat(nl), q_i.repeat(nl),
                                  g_i.repeat(nl)]).to(dev, non_blocking=True)
            dn_per_layer, _ = self._layer_losses(dn_logits, dn_boxes, tuple(sel_dn), tgt_labels, tgt_boxes,
                                                 num_boxes * scalar)
            for k, v in dn_per_layer.items():
                fam_keys.append([[nm + f"_dn_{i}" for i in range(nl - 1)] + [nm + "_dn"] for nm in k.split("|")])
                fam_vecs.append(v)
        # ONE weighted vector for all terms of this head: weight_dict per key, 1 for keys it does not name (the *_dn terms,
        # as in the reference: heads.py compute_losses only scales keys found in the dict)
        keys = [k for fams in fam_keys for ks in fams for k in ks]
        vec = torch.cat(fam_vecs) if len(fam_vecs) > 1 else fam_vecs[0]
        if weights is not None:
            w = self._weight_vector(tuple(keys), weights, vec.device)
            vec = vec * w
        out = LossDict()
        out.add_vector(keys, vec)
        return out

    def _weight_vector(self, keys, weights, device):
        cache = self.__dict__.setdefault("_wvec", {})
        ent = cache.get((keys, device))
        if ent is None:
            ent = cache[(keys, device)] = torch.tensor([float(weights.get(k, 1.0)) for k in keys], dtype=torch.float32).to(device)
        return ent
