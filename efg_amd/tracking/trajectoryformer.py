"""TrajectoryFormer training step (BASELINE configs[4]) on the MI355X path.

Counterpart of $TF/trajectoryformer.py:TrajectoryFormer ($TF = playground/tracking.3d/waymo/trajectoryformer/
trajectoryformer.centerpoint), training branch (`forward_train`, :133-242): detector boxes of the current frame and
of ten past frames -> per-frame NMS -> greedy IoU linking into trajectories -> one-step motion forecast (frozen
MotionEncoder) -> four hypotheses per track (forecast, two jittered copies, nearest detection) -> point encoder over
the LiDAR points of each hypothesis, PointNet over each hypothesis' box sequence -> global/local hypothesis encoder ->
classification and box-refinement losses.

Same constructor, module names (= checkpoint keys), input format and loss dict as the reference.  The rotated IoU /
NMS calls (hypothesis targets, linking, augmentation rejection, per-frame NMS) are this package's HIP operators
(efg_amd/operators/iou3d_nms.py).  Host-side structure is ours: batch-first attention, vectorised point crop, the
augmentation's rejection loop evaluated in one IoU launch per scene.  Behavioural quirks of the reference that change
numbers are kept and marked "(reference behaviour)".

The online tracker (`forward_inference` and its track bank, :244-438, :974-1407) lives in `online.py` and is mixed in.
"""
import copy
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..operators.iou3d_nms import boxes_iou3d_gpu, nms_gpu, nms_gpu_batched
from ..streams import side_stream
from .geometry import (crop_current_frame_points, encode_boxes_res_torch, get_corner_points_of_roi, reorder_rois,
                       rotate_points_along_z, spherical_coordinate, transform_trajs_to_global_coords,
                       transform_trajs_to_local_coords)
from .layers import (MLP, MotionEncoder, PointNet, TransformerEncoder, TransformerEncoderGlobalLocal,
                     TransformerEncoderLayer, TransformerEncoderLayerGlobalLocal)
from .losses import WeightedSmoothL1Loss, get_corner_loss
from .online import OnlineTrackingMixin

_XYZLWHR = [0, 1, 2, 3, 4, 5, -1]

# (position shift [m], size scale, heading [rad], -) per augmentation level (:457-463)
_AUG_LEVELS = ((0.5, 0.1, np.pi / 12), (0.5, 0.15, np.pi / 12), (0.5, 0.15, np.pi / 9), (0.5, 0.15, np.pi / 6),
               (0.5, 0.15, np.pi / 3))
_AUG_TRIES = 20


class TrajectoryFormer(OnlineTrackingMixin, nn.Module):
    def __init__(self, config):
        super().__init__()
        m, d = config.model, config.dataset
        self.device = torch.device(m.device)
        self.config = config
        self.is_train = config.task == "train"
        self.hidden_dim = m.hidden_dim
        self.seqboxembed = PointNet(m.boxes_dim, channels=self.hidden_dim)
        self.velboxembed = MotionEncoder(m.motion_input_dim, self.hidden_dim, out_channels=3 * m.motion_pred_frames)
        self.traj_length = d.traj_length
        self.num_lidar_points = m.num_lidar_points
        self.num_hypo_det = m.num_hypo_det
        self.num_hypo_pred = m.num_hypo_pred
        self.num_hypo_train = (self.num_hypo_pred + self.num_hypo_det) * 2
        self.num_future = m.motion_pred_frames
        self.reg_loss_func = WeightedSmoothL1Loss(code_weights=None)
        h = self.hidden_dim
        self.point_reg = MLP(h, h, 7, 3)
        self.joint_cls = MLP(h, h, 1, 3)
        self.point_cls = MLP(h, h, 1, 3)
        self.boxes_cls = MLP(h, h, 1, 3)
        self.cls_embed = MLP(h * 2 + 3, h, h, 3)
        self.up_dimension_geometry = MLP(m.point_dim, h, h, 3)
        self.dist_thresh = m.dist_thresh
        self.token = nn.Parameter(torch.zeros(1, 1, h))
        self.token_traj = nn.Parameter(torch.zeros(1, 1, h))
        self.num_encoder_layers = m.enc_layers
        self.dim_feedforward = m.dim_feedforward
        self.nhead = m.nhead
        self.encoder_fg = TransformerEncoder(
            [TransformerEncoderLayer(config, d_model=h, nhead=self.nhead, dim_feedforward=self.dim_feedforward)
             for _ in range(self.num_encoder_layers)], self.num_encoder_layers, None, config)
        self.encoder_globallocal = TransformerEncoderGlobalLocal(
            [TransformerEncoderLayerGlobalLocal(config, d_model=h, nhead=self.nhead,
                                                dim_feedforward=self.dim_feedforward)
             for _ in range(self.num_encoder_layers)], self.num_encoder_layers, None, config)
        self.train_nms_thresh = d.nms_thresh
        self.train_score_thresh = d.score_thresh
        self.load_motion_module = False
        if hasattr(m, "num_hypo_pred_eval"):      # the tracker's thresholds (eval-only keys of the reference YAML)
            self._init_online(config)
        self.to(self.device)

    # ------------------------------------------------------------------------------------------------------------
    def load_pretrain_motionencoder(self):
        """Load the `velboxembed.*` entries of the motion-prediction checkpoint and freeze the module in eval mode
        (:440-454).  Without a checkpoint path (synthetic benchmarks, tests) the current weights are kept."""
        path = getattr(self.config.dataset, "motion_model", None)
        if path:
            ckpt = torch.load(path, map_location="cpu")
            ckpt = ckpt.get("model", ckpt)
            self.velboxembed.load_state_dict({k.replace("velboxembed.", ""): v for k, v in ckpt.items()
                                              if "velboxembed" in k}, strict=True)
        self.velboxembed.eval()
        self.load_motion_module = True

    def train(self, mode=True):
        super().train(mode)
        if self.load_motion_module:
            self.velboxembed.eval()        # the forecast always uses the checkpoint's BatchNorm statistics
        return self

    def forward(self, batched_inputs):
        if not self.load_motion_module:
            self.load_pretrain_motionencoder()
        if not self.is_train:
            return self.forward_inference(batched_inputs)
        return self.forward_train(batched_inputs)

    # ------------------------------------------------------------------------------------------------------------
    def _inputs(self, batched_inputs):
        dev = self.device

        def on_device(x, dtype=None):
            t = x if torch.is_tensor(x) else torch.from_numpy(np.asarray(x))
            return t.to(device=dev, dtype=dtype, non_blocking=True)

        clouds = []
        for sample, _ in batched_inputs:
            sample = sample[0] if isinstance(sample, (list, tuple)) else sample
            clouds.append(on_device(sample["points"], torch.float32))
        annos = [info["annotations"] for _, info in batched_inputs]
        targets = [{"gt_boxes": on_device(a["gt_boxes"], torch.float32)} for a in annos]
        boxes = [on_device(a["pred_boxes3d"], torch.float32) for a in annos]
        scores = [on_device(a["pred_scores"], torch.float32) for a in annos]
        labels = [on_device(a["pred_labels"], torch.float32) for a in annos]
        return clouds, targets, boxes, scores, labels

    @staticmethod
    def _first_sample(batched_inputs):
        sample = batched_inputs[0][0]
        return sample[0] if isinstance(sample, (list, tuple)) else sample

    def prepare(self, batched_inputs):
        """The parameter-free half of the step (`_prepare`) for a batch, ahead of time: meant as the `collate` of a
        `data.loader.DeviceLoader`, i.e. on the loader's thread and HIP stream while the previous batches train.  The
        result rides on the first sample (`"prepared"`) and `forward_train` adopts it once the batch's `ready_event`
        has fired.  Runs on a shallow twin of the module (same parameters and buffers) so that the per-batch
        attributes (`batch_size`, `num_track`) of the step in flight on the main thread are not touched."""
        if not self.load_motion_module:
            self.load_pretrain_motionencoder()    # as `forward` does before its first batch
        twin = copy.copy(self)
        twin.batch_size = len(batched_inputs)
        with torch.no_grad():
            prep = twin._prepare(batched_inputs)
        self._first_sample(batched_inputs)["prepared"] = {
            "prep": prep, "batch_size": twin.batch_size, "num_track": getattr(twin, "num_track", 0)}
        return batched_inputs

    def _adopt(self, batched_inputs, prepared):
        first = self._first_sample(batched_inputs)
        main = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if main is not None:
            if first.get("ready_event") is not None:
                main.wait_event(first["ready_event"])
            else:
                torch.cuda.synchronize(self.device)   # unknown producer stream
        self.batch_size, self.num_track = prepared["batch_size"], prepared["num_track"]
        prep = prepared["prep"]
        if main is not None:
            for v in (prep or {}).values():
                for t in (v if isinstance(v, (list, tuple)) else (v,)):
                    if torch.is_tensor(t):
                        t.record_stream(main)
        return prep

    def forward_train(self, batched_inputs):
        prepared = self._first_sample(batched_inputs).get("prepared")
        if prepared is not None:
            prep = self._adopt(batched_inputs, prepared)
        else:
            self.batch_size = len(batched_inputs)
            prep = self._on_prep_stream(self._prepare, batched_inputs)
        zero = torch.zeros(1, 1, device=self.device)
        if prep is None:
            return {"loss_cls": zero, "loss_reg": zero.clone()}
        hypotheses, rois, points = prep["hypotheses"], prep["rois"], prep["points"]
        tokens = self.get_trajcetory_point_feature(hypotheses, points)
        all_tokens = torch.cat(tokens, 0)
        point_cls = self.point_cls(all_tokens).squeeze(-1)
        boxes_feat = self.get_trajectory_boxes_feature(hypotheses)
        boxes_cls = self.boxes_cls(boxes_feat).reshape(-1, self.num_hypo_train)
        hypotheses_feat = self.get_trajectory_hypotheses_feat(tokens, boxes_feat, prep["pred_labels"])
        joint_cls = torch.cat([self.joint_cls(f).squeeze(-1).reshape(-1, self.num_hypo_train)
                               for f in self.encoder_globallocal(hypotheses_feat)], 0)
        point_reg = self.point_reg(all_tokens).reshape(1, -1, 7)
        loss_cls, loss_reg = self.get_loss(rois, prep["gt_boxes"], point_cls, joint_cls, boxes_cls, point_reg,
                                           prep["ious_targets"], prep["reg_targets"], prep["fg_reg_mask"],
                                           prep["fg_iou_mask"], prep["fg_reg_idx"], prep["fg_iou_idx"])
        if prep["gt_boxes"].shape[0] > 0:
            return {"loss_cls": loss_cls, "loss_reg": loss_reg}
        return {"loss_cls": loss_cls, "loss_reg": zero}

    def _prepare(self, batched_inputs):
        """Everything of the step that involves no trainable parameter (:141-172, :204-209): NMS, linking, forecast,
        hypotheses, the point crop, targets.  Returns None for the reference's degenerate case (no track / no
        detection)."""
        clouds, targets, load_boxes3d, load_scores, load_labels = self._inputs(batched_inputs)
        pred_boxes3d, pred_labels, det_boxes3d, traj = self.organize_proposals(load_boxes3d, load_scores, load_labels)
        self.num_track = pred_boxes3d.shape[1]
        if self.num_track == 0 or det_boxes3d.shape[1] == 0:
            return None
        hypotheses_aug = self.hypotheses_augment(pred_boxes3d, targets)
        hypotheses, candidates = self.generate_trajectory_hypothses(pred_boxes3d, det_boxes3d, traj,
                                                                    self.num_hypo_det, hypotheses_aug)
        points = crop_current_frame_points(self.num_lidar_points, hypotheses, clouds)
        fg_iou_mask, fg_reg_mask, ious_targets, gt_boxes = self.get_cls_targets(pred_boxes3d, candidates, targets)
        rois = candidates[..., :7].reshape(-1, 7)
        return {"hypotheses": hypotheses, "rois": rois, "points": points, "pred_labels": pred_labels,
                "fg_iou_mask": fg_iou_mask, "fg_reg_mask": fg_reg_mask, "ious_targets": ious_targets,
                "fg_iou_idx": fg_iou_mask.nonzero().view(-1), "fg_reg_idx": fg_reg_mask.nonzero().view(-1),
                "gt_boxes": gt_boxes, "reg_targets": self.get_reg_targets(rois, gt_boxes)}

    def _on_prep_stream(self, fn, batched_inputs):
        """Run the parameter-free preparation on a side stream (GPU only, `EFG_TF_PREP_STREAM=0` disables): its ~250
        host read-backs (kept counts of 11 NMS passes per sample, boolean selections, the augmentation's accept
        flags) then wait for THAT stream only, so the host prepares step n+1 while the main stream still runs the
        backward of step n.  The side stream starts from the samples' `ready_event`s when they carry one (the point
        upload), else from the main stream's current position."""
        if self.device.type != "cuda" or os.environ.get("EFG_TF_PREP_STREAM", "1") == "0":
            return fn(batched_inputs)
        if getattr(self, "_prep_stream", None) is None:
            self._prep_stream = side_stream(self.device, "tf-prepare")
        main, side = torch.cuda.current_stream(self.device), self._prep_stream
        events = [s[0].get("ready_event") if isinstance(s, (list, tuple)) else s.get("ready_event")
                  for s, _ in batched_inputs]
        if all(e is not None for e in events):
            for e in events:
                side.wait_event(e)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            prep = fn(batched_inputs)
        main.wait_stream(side)
        for v in (prep or {}).values():
            if torch.is_tensor(v):
                v.record_stream(main)
        return prep

    # ---- proposals -> trajectories -----------------------------------------------------------------------------
    def class_agnostic_nms(self, pred_boxes3d, pred_scores, nms_thresh=0.1, score_thresh=None, nms_pre_maxsize=4096,
                           nms_post_maxsize=500):
        """Indices (into the inputs) of the boxes that survive score filtering, top-k and rotated NMS (:859-891)."""
        boxes, scores, origin = pred_boxes3d, pred_scores, None
        if score_thresh is not None:
            origin = (scores >= score_thresh).nonzero().view(-1)
            boxes, scores = boxes[origin], scores[origin]
        top_scores, order = torch.topk(scores, k=min(nms_pre_maxsize, scores.shape[0]))
        if order.shape[0] == 0:
            return torch.zeros(0, dtype=torch.long, device=boxes.device)
        keep, _ = nms_gpu(boxes[order][:, :7], top_scores, thresh=nms_thresh)
        selected = order[keep[:nms_post_maxsize]]
        return selected if origin is None else origin[selected]

    def generate_trajectory(self, proposals_list):
        """Greedy linking (:893-927): proposals_list [B, T, N, 9]; frame 0 seeds one trajectory per box, every older
        frame contributes the proposal with the highest 3-D IoU (>= 0.5) against the trajectory's last box moved
        back by 0.1 s of its velocity; a trajectory without a match holds zeros there but keeps extrapolating from
        them (reference behaviour)."""
        b, t, n, _ = proposals_list.shape
        frames = [proposals_list[:, 0]]
        valid = [torch.ones(b, n, dtype=torch.bool, device=proposals_list.device)]
        scene = torch.arange(b, device=proposals_list.device)
        for i in range(1, t):
            last, cand = frames[-1], proposals_list[:, i]
            moved = torch.cat([last[..., 0:2] - 0.1 * last[..., 6:8], last[..., 2:]], -1)
            # one IoU launch for all scenes; a scene's trajectories only see its own proposals (diagonal blocks)
            iou = boxes_iou3d_gpu(moved.reshape(b * n, 9)[:, _XYZLWHR], cand.reshape(b * n, 9)[:, _XYZLWHR])
            iou = iou.view(b, n, b, n)[scene, :, scene, :]
            best, arg = iou.max(dim=2)
            ok = best >= 0.5
            picked = torch.gather(cand, 1, arg.unsqueeze(-1).expand(-1, -1, 9))
            frames.append(torch.where(ok.unsqueeze(-1), picked, torch.zeros_like(picked)))
            valid.append(ok)
        return torch.stack(frames, 1), torch.stack(valid, 1)

    def _organize_loop(self, pred_boxes3d, pred_scores, pred_labels):
        """Per-frame NMS and padding exactly as the reference loops them (:652-690)."""
        t1 = self.traj_length + 1
        per_scene = {"box": [], "label": []}
        for boxes, scores, labels in zip(pred_boxes3d, pred_scores, pred_labels):
            boxes, scores, labels = boxes.reshape(t1, -1, 9), scores.reshape(t1, -1), labels.reshape(t1, -1)
            kept = {"box": [], "label": []}
            for j in range(t1):
                sel = self.class_agnostic_nms(boxes[j][:, [0, 1, 2, 3, 4, 5, 8]], scores[j].reshape(-1),
                                              nms_thresh=self.train_nms_thresh, score_thresh=self.train_score_thresh)
                kept["box"].append(boxes[j][sel])
                kept["label"].append(labels[j][sel].reshape(-1, 1))
            for key in kept:
                padded, _ = reorder_rois(kept[key])
                per_scene[key].append(padded.reshape(-1, padded.shape[-1]))
        frames = reorder_rois(per_scene["box"])[0].reshape(self.batch_size, t1, -1, 9)
        labels = reorder_rois(per_scene["label"])[0].reshape(self.batch_size, t1, -1, 1)
        return frames, labels

    def _organize_batched(self, pred_boxes3d, pred_scores, pred_labels, width):
        """The same result from ONE segmented NMS launch over all (scene, frame) sets and one scatter: a scene's
        frames are padded to its widest frame and flattened, the scenes padded to the longest and re-cut into
        (traj_length + 1) frames -- including the reference's frame misalignment for scenes narrower than the widest
        (reference behaviour: it pads the FLATTENED scene, :681-690)."""
        t1, b, dev = self.traj_length + 1, self.batch_size, pred_boxes3d[0].device
        boxes = pred_boxes3d[0].new_zeros(b, t1, width, 9)
        scores = pred_boxes3d[0].new_zeros(b, t1, width)
        labels = pred_boxes3d[0].new_zeros(b, t1, width)
        real = torch.zeros(b, t1, width, dtype=torch.bool, device=dev)
        for i, (bx, sc, lb) in enumerate(zip(pred_boxes3d, pred_scores, pred_labels)):
            m = bx.shape[0] // t1
            boxes[i, :, :m], scores[i, :, :m], labels[i, :, :m] = bx.reshape(t1, m, 9), sc.reshape(t1, m), lb.reshape(t1, m)
            real[i, :, :m] = True
        set_id, index, counts = nms_gpu_batched(boxes.reshape(b * t1, width, 9)[..., [0, 1, 2, 3, 4, 5, 8]],
                                                scores.reshape(b * t1, width), self.train_nms_thresh,
                                                score_thresh=self.train_score_thresh, valid=real.reshape(b * t1, width))
        count = np.asarray(counts.tolist(), np.int64).reshape(b, t1)
        per_scene = np.maximum(count.max(axis=1), 1)                     # frames of a scene padded to its widest
        length = int((per_scene * t1).max())                             # scenes padded to the longest, flattened
        start = np.concatenate([[0], np.cumsum(count.reshape(-1))[:-1]])
        start_t = torch.as_tensor(start, device=dev)
        stride_t = torch.as_tensor(np.repeat(per_scene, t1), device=dev)
        rank = torch.arange(set_id.shape[0], device=dev) - start_t[set_id]
        scene, frame = torch.div(set_id, t1, rounding_mode="floor"), set_id % t1
        dest = frame * stride_t[set_id] + rank
        frames = boxes.new_zeros(b, length, 9)
        out_labels = boxes.new_zeros(b, length, 1)
        frames[scene, dest] = boxes.reshape(b * t1, width, 9)[set_id, index]
        out_labels[scene, dest, 0] = labels.reshape(b * t1, width)[set_id, index]
        return frames.reshape(b, t1, -1, 9), out_labels.reshape(b, t1, -1, 1)

    def organize_proposals(self, pred_boxes3d, pred_scores, pred_labels):
        """Per-frame NMS, padding, linking and the one-step forecast (:652-717).  Returns the forecast boxes
        [B, N, 8], their labels [B, N, 1], the current detections [B, M, 7] and the history [B, T-1, N, 8]."""
        t1 = self.traj_length + 1
        width = max(b.shape[0] // t1 for b in pred_boxes3d)
        if width > 500:      # beyond nms_post_maxsize the per-set truncation matters: the reference's loop, literally
            frames, labels = self._organize_loop(pred_boxes3d, pred_scores, pred_labels)
        else:
            frames, labels = self._organize_batched(pred_boxes3d, pred_scores, pred_labels, width)
        det_boxes3d = frames[:, 0][..., _XYZLWHR]
        pred_vel = frames[:, 1:2][..., [6, 7]]
        traj, _ = self.generate_trajectory(frames[:, 1:])
        stamp = (torch.arange(traj.shape[1], device=traj.device) * 0.1).view(1, -1, 1, 1).expand(
            traj.shape[0], -1, traj.shape[2], 1)
        traj = torch.cat([traj[..., [0, 1, 2, 3, 4, 5, 8]], stamp], -1)
        with torch.no_grad():
            pred_hypo = self.get_pred_motion(traj, pred_vel)[:, 0, :, 0]
        return pred_hypo, labels[:, 1], det_boxes3d, traj[:, 1:]

    def get_pred_motion(self, traj, pred_vel=None, valid_steps=None):
        """Forecast of every trajectory `num_future` frames ahead (:1090-1166): constant-velocity initialisation in
        the frame of the newest box plus the MotionEncoder's (x, y, heading) residuals.  traj [B, T, N, 8],
        pred_vel [B, 1, N, 2] -> [B, num_future, N, 1, 8]; valid_steps [B] (optional): history length of each entry."""
        hist = traj.unsqueeze(3)
        b, t, n = hist.shape[:3]
        nf = self.num_future
        newest = hist[:, 0:1]
        key = (nf, traj.dtype, traj.device)   # built once: a tensor from a Python list is a blocking upload (stream sync)
        cache = self.__dict__.setdefault("_future_scale", {})
        if key not in cache:
            cache[key] = torch.tensor([0.1 * (i + 1) for i in range(nf)], dtype=traj.dtype, device=traj.device)
        scale = cache[key]
        init = newest.repeat(1, nf, 1, 1, 1)
        init[..., :2] = newest[..., :2] + scale.view(1, nf, 1, 1, 1) * pred_vel[:, 0].unsqueeze(2).unsqueeze(1)
        vel = 0.1 * pred_vel.unsqueeze(3).repeat(1, t, 1, 1, 1)
        empty = (newest[..., 3:6].sum(-1) == 0).repeat(1, t, 1, 1)
        stamp = (torch.arange(t, device=traj.device) * 0.1).view(1, t, 1, 1, 1).expand(b, -1, n, 1, 1)
        hist = torch.cat([hist, stamp.to(hist.dtype)], -1)
        centre, heading = hist[:, 0:1, :, :, 0:2], hist[:, 0:1, :, :, 6]
        hist_local, vel_local = transform_trajs_to_local_coords(hist, centre, heading, pred_vel_hypo=vel,
                                                                heading_index=6)
        init_local, _ = transform_trajs_to_local_coords(init, centre, heading, heading_index=6)
        lines = torch.cat([hist_local[..., :2], hist_local[..., 6:7], vel_local, hist_local[..., 7:8]], -1)
        lines = lines.permute(0, 2, 3, 1, 4).reshape(b, n, t, -1)
        mask = ~empty.permute(0, 2, 3, 1).reshape(b, n, t)
        if valid_steps is not None:   # [B]: batch entry i only has its first valid_steps[i] history frames (online.py)
            mask = mask & (torch.arange(t, device=traj.device).view(1, 1, t) < valid_steps.view(b, 1, 1))
        delta = self.velboxembed(lines, mask).reshape(b, n, 1, nf, 3).permute(0, 3, 1, 2, 4)
        future = init_local.clone()
        future[..., [0, 1, 6]] = delta + init_local[..., [0, 1, 6]].detach()
        return transform_trajs_to_global_coords(future, hist[:, 0:1, :, 0:1, 0:2], hist[:, 0:1, :, 0:1, 6],
                                                heading_index=6)[0]

    # ---- hypotheses ------------------------------------------------------------------------------------------------
    def hypotheses_augment(self, batch_bbox, targets):
        """Two jittered copies of every forecast box (:456-524).  A jitter (one shift / scale / rotation for the whole
        scene) is accepted when the mean over tracks of the best IoU with the ground truth stays below 0.5; the
        reference tries up to 20 draws and stops at two accepted.

        The draws come from NumPy's global generator in the reference's order.  All 20 candidates of a scene are
        evaluated in ONE IoU launch (one host sync per scene instead of up to 20); the generator is then rewound and
        advanced by exactly the number of draws the sequential loop would have consumed."""
        out = []
        for s in range(batch_bbox.shape[0]):
            bbox = batch_bbox[s]
            gt = targets[s]["gt_boxes"]
            state = np.random.get_state()
            jitters = [self._draw_jitter() for _ in range(_AUG_TRIES)]
            jit = torch.from_numpy(np.stack(jitters)).to(device=bbox.device, dtype=torch.float32)     # [20, 7]
            cand = torch.cat([bbox[None, :, 0:3] + jit[:, None, 0:3], bbox[None, :, 3:6] * jit[:, None, 3:6],
                              bbox[None, :, 6:7] + jit[:, None, 6:7]], -1)                            # [20, N, 7]
            if bbox.shape[0] > 0 and gt.shape[0] > 0:
                iou = boxes_iou3d_gpu(cand.reshape(-1, 7), gt[:, _XYZLWHR]).reshape(_AUG_TRIES, bbox.shape[0], -1)
                accepted = (iou.max(-1)[0].mean(-1) < 0.5).tolist()
            else:
                accepted = [True] * _AUG_TRIES
            chosen = [i for i, ok in enumerate(accepted) if ok][:2]
            consumed = chosen[1] + 1 if len(chosen) == 2 else _AUG_TRIES
            np.random.set_state(state)
            for _ in range(consumed):
                self._draw_jitter()
            picks = [cand[i][:, None, :] for i in chosen] + [bbox[:, None, :7]] * (2 - len(chosen))
            out.append(torch.cat(picks, 1)[None])
        return torch.cat(out)

    @staticmethod
    def _draw_jitter():
        """One augmentation draw, in the reference's generator order (:470-490): level, 3 shifts, 3 scales, 1 angle."""
        level = _AUG_LEVELS[np.random.randint(low=0, high=len(_AUG_LEVELS), size=(1,))[0]]
        shift = ((np.random.rand(3) - 0.5) / 0.5) * level[0]
        scale = ((np.random.rand(3) - 0.5) / 0.5) * level[1] + 1.0
        angle = ((np.random.rand(1) - 0.5) / 0.5) * level[2]
        return np.concatenate([shift, scale, angle])

    def generate_trajectory_hypothses(self, transfered_det, det_boxes3d, traj, num_hypo_det, aug_hypo=None):
        """Hypothesis set of every track (:719-756): [forecast | augmented copies | nearest `num_hypo_det` detections
        within dist_thresh (else a zero box)], each followed by the track's shared history.
        Returns ([B, T, N, H, 8], [B, 1, N, H, 8])."""
        b, n = transfered_det.shape[:2]
        m = det_boxes3d.shape[1]
        dist = torch.cdist(transfered_det[:, :, :2], det_boxes3d[:, :, :2], 2)
        neg, nearest = torch.topk(-dist, num_hypo_det, -1)
        matched = torch.where(-neg < self.dist_thresh, nearest, torch.full_like(nearest, m))
        with_bg = torch.cat([det_boxes3d, det_boxes3d.new_zeros(b, 1, 7)], 1)
        group = torch.gather(with_bg, 1, matched.reshape(b, -1, 1).expand(-1, -1, 7)).reshape(b, n, num_hypo_det, 7)
        group = F.pad(group, (0, 1))                                      # time stamp 0
        parts = [transfered_det[:, None, :, None, :]]
        if aug_hypo is not None:
            parts.append(F.pad(aug_hypo, (0, 1))[:, None])
        parts.append(group.unsqueeze(1))
        candidates = torch.cat(parts, 3)
        history = traj.unsqueeze(3).expand(-1, -1, -1, candidates.shape[3], -1)
        return torch.cat([candidates, history], 1), candidates

    # ---- features ----------------------------------------------------------------------------------------------------
    def get_proposal_aware_point_feature(self, src, trajectory_rois, num_rois):
        """src [B*R, T*K, 6] points, trajectory_rois [B, T, R, 8] -> [B*R, T*K, hidden] (:526-570): every point as
        offsets to its box's 8 corners and centre in spherical form (range relative to the box diagonal), plus the
        point's three feature channels."""
        k = self.num_lidar_points
        blocks = []
        for i in range(trajectory_rois.shape[1]):
            rois = trajectory_rois[:, i].reshape(self.batch_size * num_rois, -1)
            corners, _ = get_corner_points_of_roi(rois)
            anchors = torch.cat([corners.reshape(rois.shape[0], -1), rois[:, :3]], dim=-1)          # [B*R, 27]
            offsets = src[:, i * k:(i + 1) * k, :3].repeat(1, 1, 9) - anchors.unsqueeze(1)
            lwh = rois[:, 3:6]
            diag = ((lwh[:, 0] ** 2 + lwh[:, 1] ** 2 + lwh[:, 2] ** 2) ** 0.5).view(-1, 1, 1)
            blocks.append(spherical_coordinate(offsets, diag_dist=diag.expand(-1, offsets.shape[1], -1)))
        feat = torch.cat([torch.cat(blocks, dim=1), src[:, :, 3:]], dim=-1)
        return self.up_dimension_geometry(feat)

    def get_trajcetory_point_feature(self, global_trajectory_hypothses, pts):
        """Summary token of each hypothesis' current-frame points after every point-encoder layer (:572-592); `pts` =
        the cropped points of `crop_current_frame_points` [B, N*H, K, 6]."""
        n_hypo = global_trajectory_hypothses.shape[-2]
        feat = self.get_proposal_aware_point_feature(
            pts.reshape(-1, pts.shape[-2], pts.shape[-1]),
            global_trajectory_hypothses[:, 0].reshape(self.batch_size, 1, -1, 8), self.num_track * n_hypo)
        feat = feat.reshape(-1, self.num_lidar_points, feat.shape[-1])
        token = self.token.expand(self.batch_size * self.num_track * self.num_hypo_train, -1, -1)
        return self.encoder_fg(token, feat)

    def get_trajectory_boxes_feature(self, traj_rois):
        """PointNet over each hypothesis' box sequence [B, T, N, H, 8] -> [B, N, H, hidden] (:594-613)."""
        b, t, n, h, c = traj_rois.shape
        empty = traj_rois[..., :6].sum(-1) == 0
        boxes = torch.cat([traj_rois[..., :6], traj_rois[..., 6:7] % (2 * np.pi), traj_rois[..., 7:]], -1)
        boxes = torch.where(empty.unsqueeze(-1), torch.zeros_like(boxes), boxes)
        feat, _ = self.seqboxembed(boxes.permute(0, 2, 3, 4, 1).reshape(-1, c, t))
        return feat.reshape(b, n, h, feat.shape[-1])

    def get_trajectory_hypotheses_feat(self, point_feat_list, boxes_feat, pred_labels):
        """[point token | box-sequence feature | class one-hot] -> MLP -> ReLU (:615-632).  The one-hot is all zero:
        the reference writes it into a temporary produced by advanced indexing (`src[mask][..., -3:] = ...`), so
        the class never reaches the embedding (reference behaviour; checkpoints are trained that way)."""
        point_feat = point_feat_list[-1].reshape(self.batch_size, self.num_track, self.num_hypo_train, -1)
        src = torch.cat([point_feat, boxes_feat, point_feat.new_zeros(point_feat.shape[:-1] + (3,))], -1)
        return F.relu(self.cls_embed(src))

    # ---- targets and losses ------------------------------------------------------------------------------------
    @staticmethod
    def get_iou_labels(cls_iou, iou_bg_thresh=0.25, iou_fg_thresh=0.75):
        """IoU -> soft label: 0 below 0.25, 1 above 0.75, linear in between (:847-857)."""
        ramp = (cls_iou - iou_bg_thresh) / (iou_fg_thresh - iou_bg_thresh)
        return torch.where(cls_iou > iou_fg_thresh, torch.ones_like(ramp),
                           torch.where(cls_iou < iou_bg_thresh, torch.zeros_like(ramp), ramp))

    def get_cls_targets(self, pred_boxes3d, global_candidates, targets):
        """Per scene (:758-813): a track is foreground when its forecast overlaps a ground-truth box by > 0.5; every
        hypothesis is labelled with its IoU against THAT box; each hypothesis regresses to its own best box."""
        n, h = pred_boxes3d.shape[1], self.num_hypo_train
        fg, soft, reg, gts = [], [], [], []
        dev = pred_boxes3d.device
        for s in range(pred_boxes3d.shape[0]):
            gt = targets[s]["gt_boxes"]
            if n > 0 and gt.shape[0] > 0:
                gt7 = gt[:, _XYZLWHR]
                iou = boxes_iou3d_gpu(global_candidates[s][..., :7].reshape(-1, 7), gt7).reshape(n, h, -1)
                best, owner = iou[:, 0].max(-1)
                fg.append(best > 0.5)
                reg.append(iou.max(-1)[0] > 0.5)
                soft.append(self.get_iou_labels(torch.gather(iou, 2, owner.view(n, 1, 1).expand(-1, h, 1)).squeeze(-1)))
                gts.append(gt7[iou.reshape(n * h, -1).max(-1)[1]])
            else:
                soft.append(torch.zeros(n, h, device=dev))
                fg.append(torch.zeros(n, dtype=torch.bool, device=dev))
                reg.append(torch.zeros(n, h, dtype=torch.bool, device=dev))
                gts.append(torch.zeros(n * h, 7, device=dev))
        layers = self.num_encoder_layers
        fg_reg_mask = torch.cat(reg).reshape(1, -1).repeat(layers, 1).reshape(-1)
        ious_targets = torch.cat(soft, 0).reshape(-1).repeat(layers)
        return torch.cat(fg), fg_reg_mask, ious_targets, torch.cat(gts)

    def get_reg_targets(self, pred_rois, gt_boxes):
        """Ground truth in each ROI's canonical frame, heading folded into (-pi/2, pi/2), residual-coded (:815-845)."""
        ry = pred_rois[:, 6] % (2 * np.pi)
        gt = torch.cat([gt_boxes[:, 0:3] - pred_rois[:, 0:3], gt_boxes[:, 3:6], (gt_boxes[:, 6] - ry)[:, None],
                        gt_boxes[:, 7:]], -1)
        gt = rotate_points_along_z(gt.unsqueeze(1), -ry).squeeze(1)
        heading = gt[:, 6] % (2 * np.pi)
        opposite = (heading > np.pi * 0.5) & (heading < np.pi * 1.5)
        heading = torch.where(opposite, (heading + np.pi) % (2 * np.pi), heading)
        heading = torch.where(heading > np.pi, heading - np.pi * 2, heading)
        heading = torch.clamp(heading, min=-np.pi / 2, max=np.pi / 2)
        gt = torch.cat([gt[:, :6], heading[:, None], gt[:, 7:]], -1)
        local = torch.cat([torch.zeros_like(pred_rois[:, 0:3]), pred_rois[:, 3:6], torch.zeros_like(pred_rois[:, 6:7]),
                           pred_rois[:, 7:]], -1)
        return encode_boxes_res_torch(gt, local).repeat(self.num_encoder_layers, 1).reshape(1, -1, 7)

    def get_loss(self, rois, gt_boxes, point_cls, joint_cls, boxes_cls, point_reg, ious_targets, reg_targets,
                 fg_reg_mask, fg_iou_mask, fg_reg_idx=None, fg_iou_idx=None):
        """(:929-972) smooth-L1 + corner loss over the hypotheses that overlap a ground-truth box; BCE of the point
        tokens (all hypotheses), of the box-sequence features and of the joint features (foreground tracks).
        The foreground selections go through index lists (the preparation makes them, `_prepare`): indexing with the
        boolean masks themselves is a `nonzero` = a device-to-host read-back per selection, nine in this forward and four
        more in its backward, each one stopping the host at that point of the stream."""
        layers = self.num_encoder_layers
        if fg_reg_idx is None:
            fg_reg_idx = fg_reg_mask.nonzero().view(-1)
        if fg_iou_idx is None:
            fg_iou_idx = fg_iou_mask.nonzero().view(-1)
        loss_reg = self.reg_loss_func(point_reg, reg_targets).index_select(1, fg_reg_idx)
        loss_reg = loss_reg.sum() / fg_reg_mask.sum().clamp(min=1)
        loss_corner = get_corner_loss(point_reg.reshape(-1, 7), rois.repeat(layers, 1), gt_boxes.repeat(layers, 1),
                                      fg_reg_idx)
        loss_point_cls = F.binary_cross_entropy(point_cls.sigmoid().reshape(-1), ious_targets)
        per_track = ious_targets[:ious_targets.shape[0] // layers].reshape(self.batch_size * self.num_track,
                                                                          self.num_hypo_train)
        loss_box_cls = F.binary_cross_entropy(boxes_cls.sigmoid().index_select(0, fg_iou_idx),
                                              per_track.index_select(0, fg_iou_idx))
        n = fg_iou_mask.shape[0]
        fg_all = torch.cat([fg_iou_idx + n * i for i in range(layers)])        # == fg_iou_mask.repeat(layers).nonzero()
        loss_joint_cls = F.binary_cross_entropy(joint_cls.sigmoid().index_select(0, fg_all),
                                                per_track.repeat(layers, 1).index_select(0, fg_all))
        return loss_joint_cls + loss_point_cls + loss_box_cls, loss_reg + loss_corner
