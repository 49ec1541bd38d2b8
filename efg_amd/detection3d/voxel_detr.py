"""VoxelDETR / ConQueR model ($CQ/voxel_detr.py:17-291, $VD/voxel_detr.py:17-214), MI355X path.

One class serves both experiments of playground/detection.3d/waymo/conquer: a config WITH `model.contrastive` /
`model.dn` builds ConQueR (momentum GT decoder, denoising queries, contrastive query loss); a config WITHOUT them
builds plain Voxel-DETR ($VD: no `decoder_gt`, `projector`, `predictor` parameters, 17 loss terms, top-300
inference) -- configs/voxeldetr_waymo_res18.yaml.

forward(batched_inputs) keeps the reference contract: a list of `(sample, {"annotations": ...})`
pairs in, a dict of losses out (training) or per-scene detections (eval).  Differences, all on
OUR side of the operator boundary and none changing a result:
  * a sample may carry raw `points` [N,F] instead of CPU-voxelized arrays; then voxelization runs
    on the GPU for the whole batch in one call (csrc/voxelize.hip) with the per-voxel mean fused
    (the reference voxelizes with numba in DataLoader workers, extend_3d.py:255-283);
  * branches of the graph that feed nothing (FPN p2/p4-output/p5, res2_out) are not evaluated
    unless `config.model.get("eval_unused_levels")` is set (the reference computes
    and discards them; their parameters get no gradient there either);
  * the contrastive loss's Python double loop (:234-253) is evaluated in batched tensor form.
"""
import contextlib
import copy
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd.profiler import record_function

from ..modeling.backbones.fpn import build_resnet_fpn_backbone
from ..modeling.common import Conv2d
from ..modeling.readers import VoxelMeanFeatureExtractor
from ..operators import groupnorm, voxelize_batch
from ..operators.voxelize import wait_for_points
from ..operators.linear import Linear, linear
from ..spconv import core as spconv_core
from .box_coder import VoxelBoxCoder3D
from .cdn import dn_attn_mask, dn_post_process, prepare_for_cdn
from .heads import Det3DHead
from .losses import match_together, LossDict, PaddedTargets
from .position_encoding import build_position_encoding
from .transformer import Transformer


class Backbone3d(nn.Module):
    """reader -> sparse ResNet + FPN -> (feature, sine position embedding) per requested level
    ($CQ/modules/backbone3d.py:6-34)."""

    def __init__(self, hidden_dim, reader, extractor, position_encoding, out_features=()):
        super().__init__()
        self.reader = reader
        self.extractor = extractor
        self.position_encoding = build_position_encoding(position_encoding, hidden_dim)
        self.out_features = list(out_features)
        self.num_channels = [extractor.out_channels] * len(self.out_features)

    def forward(self, voxels, coordinates, num_points_per_voxel, batch_size, input_shape, voxel_mean=None):
        encoded = voxel_mean if voxel_mean is not None else self.reader(voxels, num_points_per_voxel, coordinates)
        feats = self.extractor(encoded, coordinates, batch_size, input_shape)
        return [(feats[of], self.position_encoding(feats[of]).type_as(feats[of])) for of in self.out_features]


def collate(samples, device):
    """efg/data/datasets/waymo/waymo.py:143-183 for the keys the model reads: concatenate voxels /
    counts, left-pad coordinates with the sample index."""
    voxels = torch.as_tensor(np.concatenate([s["voxels"] for s in samples], 0)).to(device)
    npv = torch.as_tensor(np.concatenate([s["num_points_per_voxel"] for s in samples], 0)).to(device)
    coors = [np.pad(s["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=i)
             for i, s in enumerate(samples)]
    coors = torch.as_tensor(np.concatenate(coors, 0)).to(device)
    return {"voxels": voxels, "num_points_per_voxel": npv, "coordinates": coors,
            "shape": np.stack([s["shape"] for s in samples], 0)}


class VoxelDETR(nn.Module):
    # set by engine.Trainer for the bucketed (overlapped) gradient exchange: callable(key, activation); see forward()
    grad_watch = None
    plain_loss_dict = False   # True: forward returns a plain dict of the scalar loss terms (torch DDP's output traversal)

    def __init__(self, config):
        super().__init__()
        self.device = torch.device(config.model.device)
        self._geo_stream = None
        self.hidden_dim = config.model.hidden_dim
        self.aux_loss = config.model.aux_loss
        self.num_classes = len(config.dataset.classes)
        self.num_queries = config.model.transformer.num_queries
        input_dim = len(config.dataset.format) if config.dataset.nsweeps == 1 else len(config.dataset.format) + 1
        self.input_dim = input_dim
        reader = VoxelMeanFeatureExtractor(**config.model.backbone.reader, num_input_features=input_dim)
        extractor = build_resnet_fpn_backbone(config.model.backbone.extractor, input_dim)
        self.backbone = Backbone3d(config.model.backbone.hidden_dim, reader, extractor,
                                   config.model.backbone.position_encoding,
                                   out_features=config.model.backbone.out_features)
        if not config.model.get("eval_unused_levels", False):
            extractor.set_active_levels(list(config.model.backbone.out_features))
        in_channels = self.backbone.num_channels
        self.input_proj = nn.ModuleList([
            nn.Sequential(Conv2d(in_channels[i], self.hidden_dim, kernel_size=1), nn.GroupNorm(32, self.hidden_dim))
            for i in range(len(self.backbone.out_features))])
        for module in self.input_proj.modules():
            if isinstance(module, nn.Conv2d):
                nn.init.xavier_uniform_(module.weight, gain=1)
                nn.init.constant_(module.bias, 0)
        tc = config.model.transformer
        self.transformer = Transformer(d_model=tc.hidden_dim, nhead=tc.nhead,
                                       nlevel=len(config.model.backbone.out_features),
                                       num_encoder_layers=tc.enc_layers, num_decoder_layers=tc.dec_layers,
                                       dim_feedforward=tc.dim_feedforward, dropout=tc.dropout,
                                       num_queries=tc.num_queries, num_classes=self.num_classes,
                                       mom=config.model.contrastive.mom if "contrastive" in config.model else None)
        self.transformer.proposal_head = Det3DHead(config, with_aux=False, with_metrics=False, num_classes=1,
                                                   num_layers=1)
        self.transformer.decoder.detection_head = Det3DHead(config, with_aux=True, with_metrics=True,
                                                            num_classes=self.num_classes, num_layers=tc.dec_layers)
        self.is_conquer = "contrastive" in config.model  # else: plain Voxel-DETR ($VD/voxel_detr.py)
        if self.is_conquer:
            # momentum ("GT") decoder: a frozen copy updated by EMA every step (voxel_detr.py:86-89)
            self.transformer.decoder_gt = copy.deepcopy(self.transformer.decoder)
            for p in self.transformer.decoder_gt.parameters():
                p.requires_grad = False
        self.box_coder = VoxelBoxCoder3D(config.dataset.voxel_size, config.dataset.pc_range, device=self.device)
        self._host_coder = VoxelBoxCoder3D(config.dataset.voxel_size, config.dataset.pc_range)
        if self.is_conquer:
            cc = config.model.contrastive
            self.eqco, self.tau, self.contras_loss_coeff = cc.eqco, cc.tau, cc.loss_coeff
            self.projector = nn.Sequential(Linear(10, cc.dim), nn.ReLU(), Linear(cc.dim, cc.dim))
            self.predictor = nn.Sequential(Linear(cc.dim, cc.dim), nn.ReLU(), Linear(cc.dim, cc.dim))
            self.similarity_f = nn.CosineSimilarity(dim=2)
        self.config = config
        vz = config.dataset.processors
        self._vox_cfg = {k: vz[k].Voxelization for k in vz if "Voxelization" in vz[k]} if isinstance(vz, dict) else {}
        pr = torch.tensor(config.dataset.pc_range, dtype=torch.float32)
        vs = torch.tensor(config.dataset.voxel_size, dtype=torch.float32)
        self.grid_size = torch.round((pr[3:] - pr[:3]) / vs).long().tolist()  # (x, y, z), voxel_generator.py:12-13
        self.noise_generator = None  # optional torch.Generator for the CDN noise (tests)
        self.to(self.device)

    # ---------------------------------------------------------------------------------------------
    def _inputs(self, batched_inputs):
        samples = [bi[0] for bi in batched_inputs]
        if "voxels" in samples[0]:  # reference format: CPU-voxelized arrays
            c = collate(samples, self.device)
            geo = self._geometry_stream()
            if geo is not None:
                # the uploads (and any dtype conversion of the coordinates) are queued on the main stream; the
                # sparse-conv geometry kernels read them on the geometry stream
                geo.wait_stream(torch.cuda.current_stream())
            return c["voxels"], c["coordinates"], c["num_points_per_voxel"], list(c["shape"][0]), None
        mode = "train" if self.training else "val"
        vc = self._vox_cfg[mode]
        pts = [torch.as_tensor(s["points"], dtype=torch.float32).to(self.device, non_blocking=True) for s in samples]
        geo = self._geometry_stream()
        if geo is None:
            out = voxelize_batch(pts, vc.voxel_size, vc.pc_range, vc.max_points_in_voxel, vc.max_voxel_num)
            mean = out["voxel_mean"][:, : self.input_dim].contiguous()
        else:
            # voxelization sizes every downstream tensor (one count readback): run it on the geometry stream so
            # the readback does not wait for the previous step's backward still queued on the main stream
            main = torch.cuda.current_stream()
            wait_for_points(geo, main, samples, pts)
            with torch.cuda.stream(geo):
                out = voxelize_batch(pts, vc.voxel_size, vc.pc_range, vc.max_points_in_voxel, vc.max_voxel_num)
                mean = out["voxel_mean"][:, : self.input_dim].contiguous()
            main.wait_stream(geo)
            for t in (out["voxels"], out["coordinates"], out["num_points_per_voxel"], mean):
                t.record_stream(main)
        return out["voxels"], out["coordinates"], out["num_points_per_voxel"], self.grid_size, mean

    @staticmethod
    def _project(proj, x):
        """input_proj[i](x) = GroupNorm(Conv2d 1x1 (x)) (:43-51).  On the GPU both run on the channels-last map: the
        convolution as a GEMM, the norm by the HIP kernel in the same layout -- which is also the [B, H*W, C] token
        layout the encoder reads, so no transposing copy is left between the backbone and the transformer."""
        conv, norm = proj[0], proj[1]
        if (len(proj) == 2 and x.is_cuda and isinstance(conv, Conv2d) and conv._is_pointwise() and conv.norm is None
                and conv.activation is None):
            y = linear(x.permute(0, 2, 3, 1), conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)
            if groupnorm.fusable(y, norm):
                return groupnorm.group_norm_nhwc(y, norm).permute(0, 3, 1, 2)
            return norm(y.permute(0, 3, 1, 2))
        return proj(x)

    def _geometry_stream(self):
        """High-priority side stream for voxelization + sparse-conv geometry (spconv/core.py `geometry_stream`);
        None on CPU or with EFG_GEOMETRY_STREAM=0."""
        if self.device.type != "cuda" or os.environ.get("EFG_GEOMETRY_STREAM", "1") == "0":
            return None
        if self._geo_stream is None:
            from ..streams import side_stream

            self._geo_stream = side_stream(self.device, "geometry", priority=-1)   # one per process and device
        return self._geo_stream

    def forward(self, batched_inputs):
        batch_size = len(batched_inputs)
        with record_function("efg::voxelize"):
            voxels, coords, num_points_per_voxel, input_shape, voxel_mean = self._inputs(batched_inputs)
        if self.training:
            # annotations arrive as host arrays: normalise them on the host (box_coder.encode is ~25 tiny
            # kernels and one sync per scene on the device) and upload the encoded targets
            host_targets = []
            for bi in batched_inputs:
                ann = bi[1]["annotations"]
                tgt = {"gt_boxes": torch.as_tensor(np.asarray(ann["gt_boxes"])).float().clone(),
                       "labels": torch.as_tensor(np.asarray(ann["labels"])).long().clone()}
                host_targets.append(self._host_coder.encode(tgt))
            # padded batch form, built on the host and uploaded once (two small asynchronous copies); the per-scene
            # dicts the transformer / losses index are views of it
            counts = [int(t["labels"].numel()) for t in host_targets]
            g = max(max(counts), 1)
            labels = torch.zeros(batch_size, g, dtype=torch.int64)
            boxes = torch.zeros(batch_size, g, 7, dtype=torch.float32)
            for b, t in enumerate(host_targets):
                labels[b, : counts[b]] = t["labels"]
                boxes[b, : counts[b]] = t["gt_boxes"]
            targets = PaddedTargets(labels.to(self.device, non_blocking=True),
                                    boxes.to(self.device, non_blocking=True), counts)
        else:
            targets = host_targets = None
        with record_function("efg::backbone+fpn"):
            with spconv_core.geometry_stream(self._geometry_stream()):
                feats_pos = self.backbone(voxels, coords, num_points_per_voxel, batch_size, input_shape, voxel_mean)
            features = [self._project(self.input_proj[i], fp[0]) for i, fp in enumerate(feats_pos)]
        pos_encodings = [fp[1] for fp in feats_pos]
        watch = getattr(self, "grad_watch", None)
        if watch is not None and self.training:
            # data-parallel exchange in buckets (engine.BucketedGradientAllReduce): the gradient of these activations
            # marks the moment backward has finished the transformer, resp. the neck
            for f in features:
                watch("transformer", f)
            for t in getattr(self.backbone.extractor, "last_bottom_up", {}).values():
                watch("neck", t)
        dn = self.config.model.dn if self.is_conquer else None
        if self.training and dn is not None and dn.enabled and dn.dn_number > 0:
            with record_function("efg::cdn"):
                # the denoising queries are a few hundred numbers derived from host annotations: build them on
                # the host (CPU generator -> device-independent noise) and upload three small tensors
                input_query_label, input_query_bbox, _, dn_meta = prepare_for_cdn(
                    dn_args=(host_targets, dn.dn_number, dn.dn_label_noise_ratio, dn.dn_box_noise_scale),
                    training=self.training, num_queries=self.num_queries, num_classes=self.num_classes,
                    hidden_dim=self.hidden_dim, label_enc=None, generator=self.noise_generator, with_mask=False)
                input_query_label = input_query_label.to(self.device, non_blocking=True)
                input_query_bbox = input_query_bbox.to(self.device, non_blocking=True)
                attn_mask = dn_attn_mask(dn_meta["pad_size"], dn_meta["pad_size"] // (2 * dn.dn_number),
                                         dn.dn_number, self.num_queries, self.device)
        else:
            input_query_bbox = input_query_label = attn_mask = dn_meta = None
        with record_function("efg::transformer"):
            hidden_state, init_reference, inter_references, src_embed, src_ref_windows, src_indexes = self.transformer(
                features, pos_encodings, input_query_bbox, input_query_label, attn_mask,
                targets=targets if self.is_conquer else None)
        head = self.transformer.decoder.detection_head
        outputs_classes, outputs_coords = [], []
        for idx in range(hidden_state.shape[0]):
            reference = init_reference if idx == 0 else inter_references[idx - 1]
            oc, ob = head(hidden_state[idx], reference, idx)
            outputs_classes.append(oc)
            outputs_coords.append(ob)
        outputs_class, outputs_coord = torch.stack(outputs_classes), torch.stack(outputs_coords)
        if dn is not None and dn.dn_number > 0 and dn_meta is not None:
            outputs_class, outputs_coord = dn_post_process(outputs_class, outputs_coord, dn_meta, self.aux_loss,
                                                           self._set_aux_loss)
        if not self.training:
            return self._inference(outputs_class, outputs_coord)
        with record_function("efg::losses"):
            losses = self._losses(outputs_class, outputs_coord, targets, dn_meta, src_embed, src_ref_windows, src_indexes)
            return dict(losses) if self.plain_loss_dict else losses   # (torch DDP wrappers: engine.Trainer)

    def _losses(self, outputs_class, outputs_coord, targets, dn_meta, src_embed, src_ref_windows, src_indexes):
        head = self.transformer.decoder.detection_head
        losses = LossDict()
        # encoder proposal losses (class-agnostic), voxel_detr.py:198-209
        bin_targets = targets.class_agnostic() if isinstance(targets, PaddedTargets) else copy.deepcopy(targets)
        if not isinstance(targets, PaddedTargets):
            for tgt in bin_targets:
                tgt["labels"].fill_(0)
        enc_outputs = dict(self.transformer.enc_outputs)  # class logits of all tokens + boxes of the top-k (one evaluation)
        nq = self.num_queries
        outputs = {"pred_logits": outputs_class[-1][:, :nq], "pred_boxes": outputs_coord[-1][:, :nq],
                   "aux_outputs": self._set_aux_loss(outputs_class[:-1, :, :nq], outputs_coord[:-1, :, :nq])}
        # the encoder-proposal and the decoder matchings in ONE assignment launch (losses.match_together); host tensors:
        # every loss matches for itself, as the reference does
        phead = self.transformer.proposal_head
        prep_e = prep_d = q_e = q_d = sim = None
        if outputs_class.is_cuda:
            prep_e, prep_d = phead.losses.prepare(enc_outputs, bin_targets), head.losses.prepare(outputs, targets)
            # (on the main stream: the assignment on the shared side stream beside the matching-independent half of the
            # contrastive loss measured neutral in round 4 -- 31.69 / 31.74 against 31.68 / 31.76 -- and 61-65 ms per step on a
            # stream of its own, streams.py; the switch was retired in round 6)
            both = match_together([(phead.losses.matcher, prep_e), (head.losses.matcher, prep_d)])
            if self.is_conquer and dn_meta is not None and sum(t["gt_boxes"].shape[0] for t in targets) > 0:
                sim = self._contrastive_similarity(outputs_class, outputs_coord)
            if both is None:
                prep_e = prep_d = None
            else:
                q_e, q_d = both
        enc_losses = phead.compute_losses(enc_outputs, bin_targets, prepared=prep_e, q_of_g=q_e)
        losses.merge(enc_losses, "_enc")
        with record_function("efg::losses.decoder"):
            losses.merge(head.compute_losses(outputs, targets, dn_meta, prepared=prep_d, q_of_g=q_d))
        if self.is_conquer:
            with record_function("efg::losses.contrastive"):
                losses.merge(self._contrastive_losses(outputs_class, outputs_coord, outputs["matched_query_of_gt"],
                                                      targets, dn_meta, sim=sim))
        return losses

    def _contrastive_similarity(self, outputs_class, outputs_coord):
        """The half of the contrastive loss that does not depend on the matching: projections of the noised-GT rows and of the
        queries, normalised, and EVERY noised-GT row against every query of its scene in one [R, C] x [C, Q] product per
        (layer, scene) -> [L, B, R, Q] / tau.  (Gathering the per-pair operands first, as the loop form suggests,
        materialises a [L, n, Q, C] copy of the query projections -- 245 MB for 80 boxes -- and runs L*n small products over
        it, forward and backward.)  `_losses` issues it while the assignment kernel runs on the matching stream."""
        nq, n_layers = self.num_queries, self.config.model.transformer.dec_layers
        projs = torch.cat((outputs_class[:n_layers], outputs_coord[:n_layers]), dim=-1)     # [L, B, Q+gt, 10]
        gt_projs = self.projector(projs[:, :, nq:].detach())                                 # [L, B, gt, C]
        pred_projs = self.predictor(self.projector(projs[:, :, :nq]))                        # [L, B, Q, C]
        gt_n = F.normalize(gt_projs, dim=-1, eps=1e-8)                                        # [L, B, R, C]
        pn = F.normalize(pred_projs, dim=-1, eps=1e-8)                                        # [L, B, Q, C]
        return torch.matmul(gt_n, pn.transpose(-1, -2)) / self.tau

    def _contrastive_losses(self, outputs_class, outputs_coord, query_of_gt, targets, dn_meta, sim=None):
        """voxel_detr.py:223-254 in batched form.  For decoder layer li and scene bi, every matched
        (query p, gt g) contributes the mean over the G positive-noised GT copies r = g + max_gt*pi of
        log(exp(s[r,p]) + sum_{q unmatched} exp(s[r,q])) - s[r,p], s = cos-sim / tau.  All layers and
        scenes are evaluated together (the reference runs a Python loop per layer, scene and pair).
        query_of_gt: int64 [B, G] on the device, the last layer's assignment (matcher.match_layers)."""
        out = LossDict()
        per_gt = [t["gt_boxes"].shape[0] for t in targets]
        max_gt, num_gts = max(per_gt), sum(per_gt)
        if num_gts == 0 or dn_meta is None:
            return out
        nq, groups = self.num_queries, dn_meta["num_dn_group"]
        dev = outputs_class.device
        n_layers = self.config.model.transformer.dec_layers
        if sim is None:
            sim = self._contrastive_similarity(outputs_class, outputs_coord)
        # (scene, gt) of all matched pairs follow from the GT counts (host -> one asynchronous upload); the
        # matched query stays on the device
        static = torch.stack([torch.cat([torch.full((n,), bi, dtype=torch.int64) for bi, n in enumerate(per_gt)]),
                              torch.cat([torch.arange(n, dtype=torch.int64) for n in per_gt])]).to(
                                  dev, non_blocking=True)
        b_idx, g_idx = static[0], static[1]
        q_idx = query_of_gt[b_idx, g_idx].clamp(min=0)  # -1 = unmatched (infeasible assignment; the engine raises)
        n = num_gts
        neg_mask = torch.ones(len(targets), nq, dtype=torch.bool, device=dev)
        neg_mask.index_put_((b_idx, q_idx), torch.zeros((), dtype=torch.bool, device=dev))  # unmatched queries
        rows = g_idx[:, None] + (torch.arange(1, groups + 1, device=dev) * max_gt)[None, :]  # [n, G]
        sim = sim[:, b_idx[:, None], rows]                                                    # [L, n, G, Q]
        pos = sim.gather(3, q_idx[None, :, None, None].expand(n_layers, n, groups, 1))
        neg = (torch.exp(sim) * neg_mask[b_idx][None, :, None, :]).sum(dim=-1, keepdim=True)
        per_layer = (torch.log(torch.exp(pos) + neg) - pos).mean(dim=(2, 3)).sum(dim=1)      # [L]
        out.add_vector([f"loss_contrastive_dec_{li}" for li in range(n_layers)],
                       per_layer * (self.contras_loss_coeff / num_gts))
        return out

    def _inference(self, outputs_class, outputs_coord):
        """ConQueR ($CQ/voxel_detr.py:257-284): keep every (query, class) with score >= 0.1 (batch 1);
        Voxel-DETR ($VD/voxel_detr.py:166-200): the 300 best (query, class) pairs per scene (unsorted top-k)."""
        out_logits = outputs_class[-1][:, : self.num_queries]
        out_bbox = outputs_coord[-1][:, : self.num_queries]
        out_prob = out_logits.sigmoid().view(out_logits.shape[0], -1)
        out_bbox = self.box_coder.decode(out_bbox.clone())
        if not self.is_conquer:
            ncls = out_logits.shape[2]
            scores, keep = torch.topk(out_prob, min(300, out_prob.shape[1]), dim=1, sorted=False)
            box_idx = keep.div(ncls, rounding_mode="floor")
            labels = keep % ncls + 1
            boxes = torch.gather(out_bbox, 1, box_idx.unsqueeze(-1).repeat(1, 1, out_bbox.shape[-1]))
            return [{"scores": s.detach().cpu(), "labels": l.detach().cpu(), "boxes3d": b.detach().cpu()}
                    for s, l, b in zip(scores, labels, boxes)]
        keep = torch.nonzero(out_prob >= 0.1, as_tuple=True)[1]
        scores = out_prob[:, keep]
        ncls = out_logits.shape[2]
        box_idx = keep.view(1, -1).div(ncls, rounding_mode="floor")
        labels = keep.view(1, -1) % ncls + 1
        boxes = torch.gather(out_bbox, 1, box_idx.unsqueeze(-1).repeat(1, 1, out_bbox.shape[-1]))
        return [{"scores": s.detach().cpu(), "labels": l.detach().cpu(), "boxes3d": b.detach().cpu()}
                for s, l, b in zip(scores, labels, boxes)]

    @staticmethod
    def _set_aux_loss(outputs_class, outputs_coord):
        return [{"pred_logits": a, "pred_boxes": b} for a, b in zip(outputs_class, outputs_coord)]
