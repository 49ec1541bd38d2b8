"""TrajectoryFormer's online tracker (inference): one frame at a time, batch 1.

Counterpart of $TF/trajectoryformer.py:forward_inference (:244-408) and its helpers (`init_trajectory` :974-1037,
`get_history_traj` :410-438, `get_point_and_trajectory` :1039-1073, `get_pred_candi` :1075-1088, `get_det_candi`
:1168-1220, `genereate_trajcetory_hypotheses_inference` :1222-1250, `generate_refined_boxes` :1252-1268,
`get_keep_mask` :1270-1284, `update_trajectory` :1286-1383) and of $TF/modules/tracker.py (the CenterPoint-style
greedy centre-distance association that proposes the detection hypothesis of every track).

Same inputs (`batched_inputs` of one `(sample, info)` pair; `info["token"]` carries the frame number,
`info["veh_to_global"]` the pose, `info["annotations"]` the detector's boxes / scores / labels) and the same result
(`[{"track_scores", "track_labels", "track_boxes3d", "track_ids"}]`) as the reference.  State is ours: a bank of
per-track global-frame histories held as NumPy rows (the reference keeps Python lists of 0-d device tensors and a list
of per-frame `Instances`), the association on arrays instead of lists of dicts, the history of all tracks moved into
the current vehicle frame in one product.  The learned part runs through the same modules as the training step.
"""
import collections

import os

import numpy as np
import torch

from ..operators.iou3d_nms import boxes_iou3d_gpu
from .geometry import crop_current_frame_points, decode_torch, rotate_points_along_z

CLASS_NAMES = ("VEHICLE", "PEDESTRIAN", "CYCLIST")


def boxes_to_global(boxes, vels, pose):
    """Vehicle-frame boxes [n, 7] / velocities [n, 2] (NumPy) -> global frame, float64 arithmetic then fp32, as
    `transform_box_to_global` ($TF/modules/utils.py:434-455)."""
    n = boxes.shape[0]
    xyz1 = np.concatenate([boxes[:, :3], np.ones((n, 1))], axis=-1)
    vel0 = np.concatenate([vels[:, 0:2], np.zeros((n, 1))], axis=-1)
    centre = np.dot(xyz1, pose.T)[:, :3]
    out = np.concatenate([centre, boxes[:, 3:7]], axis=-1)
    out[:, -1] = out[:, -1] + np.arctan2(pose[1, 0], pose[0, 0])
    return out.astype(np.float32), np.dot(vel0, pose[:3, :3].T)[:, :2].astype(np.float32)


class GreedyCentreTracker:
    """$TF/modules/tracker.py:PubTracker on arrays.  `reset` installs the tracks of the last frame (their index is
    their id); `associate` pushes every detection back by its velocity x time lag and matches detections, in order,
    to the nearest free track of the same class within the class's distance gate."""

    def __init__(self, max_dist):
        self.max_dist = dict(max_dist)
        self.centres = np.zeros((0, 2), np.float32)
        self.labels = np.zeros((0,), np.int32)

    def reset(self, centres, labels):
        self.centres = np.asarray(centres, np.float32).reshape(-1, 2)
        self.labels = np.asarray(labels, np.int32).reshape(-1)

    def associate(self, centres, velocities, labels, time_lag):
        """-> list of (detection index, track index)."""
        n, m = len(centres), len(self.centres)
        if n == 0:
            self.reset(np.zeros((0, 2)), np.zeros((0,)))
            return []
        if m == 0:
            return []
        labels = np.asarray(labels, np.int32)
        moved = (centres.astype(np.float32) + (velocities.astype(np.float32) * -1 * time_lag).astype(np.float32))
        gate = np.array([self.max_dist[CLASS_NAMES[int(c) - 1]] for c in labels], np.float32)
        dist = np.sqrt(((self.centres.reshape(1, -1, 2) - moved.reshape(-1, 1, 2)) ** 2).sum(axis=2))
        invalid = ((dist > gate.reshape(n, 1)) + (labels.reshape(n, 1) != self.labels.reshape(1, m))) > 0
        dist = dist + invalid * 1e18
        pairs = []
        for i in range(n):
            j = int(dist[i].argmin())
            if dist[i][j] < 1e16:
                dist[:, j] = 1e18
                pairs.append((i, j))
        return pairs


class OnlineTrackingMixin:
    """`forward_inference` for `TrajectoryFormer` (mixed into the model class)."""

    def _init_online(self, config):
        m = config.model
        self.nms_thresh = m.nms_thresh
        self.num_hypo_inference = m.num_hypo_pred_eval
        self.history_traj_frames = m.history_frames_eval
        self.keep_thresh = {1: m.track_score.car, 2: m.track_score.ped, 3: m.track_score.cyc}
        self.new_born = {1: m.new_born_score.car, 2: m.new_born_score.ped, 3: m.new_born_score.cyc}
        self.new_born_nms_thresh = m.new_born_nms_thresh
        self.eval_class = m.eval_class
        self.tracker = GreedyCentreTracker(m.max_dist)
        self.max_id = 0
        self._reset_tracks()

    def _reset_tracks(self):
        self.bank = collections.defaultdict(lambda: {"boxes": [], "vels": []})   # id -> global rows, newest first
        self.frames_seen = 0
        self.current = None        # tracks of the last frame: ids, boxes, vels, scores, labels (device tensors)

    # ---- state ---------------------------------------------------------------------------------------------------
    def _store(self, ids, boxes, refined, vels, scores, labels, pose):
        """Append this frame's tracks to the bank (global frame) and make them the association targets."""
        self.current = {"ids": ids, "boxes": boxes, "vels": vels, "scores": scores, "labels": labels}
        self.frames_seen += 1
        gbox, gvel = boxes_to_global(boxes.cpu().numpy(), vels.cpu().numpy(), pose)
        for row, tid in enumerate(ids.tolist()):
            self.bank[int(tid)]["boxes"].insert(0, gbox[row])
            self.bank[int(tid)]["vels"].insert(0, gvel[row])
        out = {"track_scores": scores.detach().cpu(), "track_labels": labels.detach().cpu(),
               "track_boxes3d": refined.detach().cpu(), "track_ids": ids.detach().cpu().int()}
        return out, gbox, gvel

    def init_trajectory(self, pose, det_boxes, det_scores, det_vels, det_labels):
        """Start over from this frame's detections (:974-1037): NMS + the evaluated class's new-born score."""
        if self.eval_class not in CLASS_NAMES:
            raise NotImplementedError("model.eval_class must be one of %s" % (CLASS_NAMES,))
        self._reset_tracks()
        keep = self.class_agnostic_nms(det_boxes, det_scores.reshape(-1), nms_thresh=self.nms_thresh,
                                       score_thresh=self.new_born[CLASS_NAMES.index(self.eval_class) + 1])
        ids = torch.arange(keep.shape[0], device=det_boxes.device)
        self.max_id = int(keep.shape[0])
        out, gbox, gvel = self._store(ids, det_boxes[keep], det_boxes[keep], det_vels[keep], det_scores[keep],
                                      det_labels[keep], pose)
        return out, gbox, gvel

    def get_history_traj(self, ids, pose):
        """The tracks' last `num_hypo_inference + history_traj_frames` boxes in the CURRENT vehicle frame (:410-438):
        [1, W, N, 7] and velocities [1, W, N, 2], zeros where a track is younger than the window."""
        span = self.num_hypo_inference + self.history_traj_frames
        window = min(self.frames_seen, span)
        n = len(ids)
        boxes = np.zeros((window, n, 7), np.float32)
        vels = np.zeros((window, n, 2), np.float32)
        have = np.zeros((window, n), bool)
        for k, tid in enumerate(ids):
            rows = self.bank[int(tid)]["boxes"][:span]
            boxes[:len(rows), k] = np.stack(rows)
            vels[:len(rows), k] = np.stack(self.bank[int(tid)]["vels"][:span])
            have[:len(rows), k] = True
        dev = self.device
        g = torch.from_numpy(boxes).to(dev).reshape(-1, 7)
        v = torch.from_numpy(vels).to(dev).reshape(-1, 2)
        global_from_ref = torch.from_numpy(np.asarray(pose)).to(dev).float()
        ref_from_global = torch.linalg.inv(global_from_ref)
        xyz1 = torch.cat([g[:, :3], torch.ones_like(g[:, :1])], -1)
        centre = torch.mm(ref_from_global, xyz1.t()).t()[:, :3]
        vel = torch.mm(ref_from_global[:3, :3], torch.cat([v, torch.zeros_like(v[:, :1])], -1).t()).t()[:, :2]
        local = torch.cat([centre, g[:, 3:7]], -1)
        local[:, 6] = local[:, 6] - torch.atan2(global_from_ref[1, 0], global_from_ref[0, 0])
        mask = torch.from_numpy(have).to(dev).reshape(-1, 1)
        traj = torch.where(mask, local, torch.zeros_like(local)).reshape(1, window, n, 7)
        return traj, torch.where(mask, vel, torch.zeros_like(vel)).reshape(1, window, n, 2)

    # ---- hypotheses ----------------------------------------------------------------------------------------------
    def get_pred_candi(self, traj, traj_vels):
        """Forecast hypotheses (:1075-1088): the motion model run from each of the last `num_pred` frames, i + 1 steps
        ahead, so that every forecast lands on the current frame."""
        num_pred = max(1, min(self.num_hypo_inference, traj.shape[1] - 1))
        h = self.history_traj_frames
        if num_pred > 1 and traj.shape[0] == 1 and os.environ.get("EFG_TRACKER_BATCH", "1") != "0":
            # the `num_pred` forecasts as ONE batch of the motion model (1.7 ms of launches per forecast otherwise): window i
            # = frames i .. i + h - 1 of the history, zero-padded to the longest; the padded steps are masked out of the
            # polyline encoder (its layers are row-wise, BatchNorm in eval mode, and its max-pools see ReLU outputs, so
            # the zero rows of masked steps change nothing)
            t = traj.shape[1]
            lens = [min(h, t - i) for i in range(num_pred)]
            wins = traj.new_zeros(num_pred, lens[0], traj.shape[2], traj.shape[3])
            for i, n_i in enumerate(lens):
                wins[i, :n_i] = traj[0, i:i + n_i]
            future = self.get_pred_motion(wins, torch.cat([traj_vels[:, i:i + 1] for i in range(num_pred)], 0),
                                          valid_steps=torch.as_tensor(lens, device=traj.device))
            hyps = [future[i:i + 1, i] for i in range(num_pred)]
        else:
            hyps = []
            for i in range(num_pred):
                future = self.get_pred_motion(traj[:, i:i + h], traj_vels[:, i:i + 1])
                hyps.append(future[:, i])
        pred = torch.cat(hyps, 2)
        empty = pred[..., 3:6].sum(-1) == 0
        return torch.where(empty.unsqueeze(-1), torch.zeros_like(pred), pred)

    def get_det_candi(self, pose, det_boxes, det_vels, det_labels, num_track):
        """The detection hypothesis of every track (:1168-1220): greedy centre association in the global frame."""
        gbox, gvel = boxes_to_global(det_boxes.cpu().numpy(), det_vels.cpu().numpy(), pose)
        pairs = self.tracker.associate(gbox[:, :2], gvel, det_labels.cpu().numpy(), time_lag=0.1)
        det_candi = det_boxes.new_zeros(1, num_track, 7)
        det_vel = det_boxes.new_zeros(1, num_track, 2)
        asso = torch.zeros(num_track, dtype=torch.bool, device=det_boxes.device)
        if pairs:
            d = torch.as_tensor([p[0] for p in pairs], device=det_boxes.device)
            t = torch.as_tensor([p[1] for p in pairs], device=det_boxes.device)
            det_candi[0, t] = det_boxes[d]
            det_vel[0, t] = det_vels[d]
            asso[t] = True
        return det_candi, det_vel, asso

    def generate_refined_boxes(self, rois, box_preds):
        """rois [N, H, 7+], residuals [N, H, 7] -> refined boxes in the vehicle frame (:1252-1268)."""
        n = rois.shape[0]
        flat = rois.reshape(-1, rois.shape[-1])
        anchors = torch.cat([torch.zeros_like(flat[:, :3]), flat[:, 3:7]], -1).detach()
        boxes = decode_torch(box_preds.reshape(-1, 7), anchors)
        boxes = rotate_points_along_z(boxes.unsqueeze(1), flat[:, 6]).squeeze(1)
        boxes = torch.cat([boxes[:, :3] + flat[:, :3], boxes[:, 3:]], -1)
        return boxes.reshape(n, -1, 7)

    # ---- one frame -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_inference(self, batched_inputs):
        assert len(batched_inputs) == 1, "the online tracker runs one frame at a time"
        self.batch_size = 1
        sample, info = batched_inputs[0]
        sample = sample[0] if isinstance(sample, (list, tuple)) else sample
        frame_id = int(str(info["token"]).split("_frame_")[-1].split(".")[0])
        pose = np.asarray(info["veh_to_global"])
        dev = self.device
        ann = info["annotations"]
        det9 = torch.as_tensor(np.asarray(ann["pred_boxes3d"]), dtype=torch.float32, device=dev)
        det_scores = torch.as_tensor(np.asarray(ann["pred_scores"]), dtype=torch.float32, device=dev)
        det_labels = torch.as_tensor(np.asarray(ann["pred_labels"]), dtype=torch.float32, device=dev)
        det_boxes, det_vels = det9[:, [0, 1, 2, 3, 4, 5, 8]], det9[:, [6, 7]]
        points = sample["points"]
        points = points if torch.is_tensor(points) else torch.from_numpy(np.asarray(points))
        points = points.to(device=dev, dtype=torch.float32)

        if frame_id == 0:
            out, gbox, _ = self.init_trajectory(pose, det_boxes, det_scores, det_vels, det_labels)
            self.tracker.reset(gbox[:, :2], self.current["labels"].cpu().numpy())
            return [out]

        keep = self.class_agnostic_nms(det_boxes, det_scores.reshape(-1), nms_thresh=self.nms_thresh)
        det_boxes, det_vels, det_scores, det_labels = det_boxes[keep], det_vels[keep], det_scores[keep], det_labels[keep]
        if self.current is None or self.current["ids"].shape[0] == 0:
            if bool((det_boxes.sum(-1) == 0).all()):
                return [{"track_scores": torch.zeros(0), "track_labels": torch.zeros(0),
                         "track_boxes3d": torch.zeros(0, 7), "track_ids": torch.zeros(0).int()}]
            return [self.init_trajectory(pose, det_boxes, det_scores, det_vels, det_labels)[0]]   # (tracker not reset:
            #                                                                                        reference behaviour)
        cur = self.current
        ids = cur["ids"].tolist()
        n = len(ids)
        self.num_track = n
        traj, traj_vels = self.get_history_traj(ids, pose)
        hist, hist_vels = traj[:, :self.history_traj_frames - 1], traj_vels[:, :self.history_traj_frames - 1]
        pred = self.get_pred_candi(hist, hist_vels)                                         # [1, N, P, 7]
        det_candi, det_vel, asso = self.get_det_candi(pose, det_boxes, det_vels, det_labels, n)
        stamp = ((torch.arange(hist.shape[1], device=dev) + 1) * 0.1).view(1, -1, 1, 1).expand(1, -1, n, 1)
        hist = torch.cat([hist, stamp], -1)
        candidates = torch.cat([pred, det_candi.unsqueeze(2)], 2)                           # [1, N, P + 1, 7]
        candidates = torch.cat([candidates, torch.zeros_like(candidates[..., :1])], -1).unsqueeze(1)
        n_hypo = candidates.shape[3]
        self.num_candi = n_hypo
        hypotheses = torch.cat([candidates, hist.unsqueeze(3).expand(-1, -1, -1, n_hypo, -1)], 1)
        cand_vels = torch.cat([cur["vels"][None, :, None, :].expand(-1, -1, n_hypo - 1, -1), det_vel.unsqueeze(2)], 2)
        cand_vels = cand_vels.reshape(n, n_hypo, 2)
        candidates = candidates.reshape(n, n_hypo, 8)

        pts = crop_current_frame_points(self.num_lidar_points, hypotheses, [points])
        feat = self.get_proposal_aware_point_feature(pts.reshape(-1, pts.shape[-2], pts.shape[-1]),
                                                     hypotheses[:, 0].reshape(1, 1, -1, 8), n * n_hypo)
        feat = feat.reshape(-1, self.num_lidar_points, feat.shape[-1])
        tokens = self.encoder_fg(self.token.expand(n * n_hypo, -1, -1), feat)
        fg_confidence = self.point_cls(tokens[-1]).reshape(n, n_hypo).sigmoid()
        boxes_feat = self.get_trajectory_boxes_feature(hypotheses[:, :self.history_traj_frames])
        point_feat = tokens[-1].reshape(1, n, n_hypo, -1)
        src = torch.cat([point_feat, boxes_feat, point_feat.new_zeros(1, n, n_hypo, 3)], -1)   # class one-hot stays
        joint = self.encoder_globallocal(torch.relu(self.cls_embed(src)))                      # zero, as in training
        hypo_scores = self.joint_cls(joint[-1]).reshape(-1, n_hypo).sigmoid()
        refined = self.generate_refined_boxes(candidates[..., :7], self.point_reg(tokens[-1]).reshape(n, -1, 7))

        # a track survives if a detection was associated to it, else if its forecast is confident enough for its class
        labels = cur["labels"]
        thresh = torch.zeros_like(fg_confidence[:, 0])
        for c, t in self.keep_thresh.items():
            thresh = torch.where(labels == c, torch.full_like(thresh, t), thresh)
        confident = (fg_confidence[:, 0] > thresh) & (labels >= 1) & (labels <= 3)
        keep_mask = torch.where(asso, torch.ones_like(asso), confident)
        selected = hypo_scores.max(-1)[1][keep_mask]
        rows = keep_mask.nonzero().flatten()
        matched = {"boxes": candidates[rows, selected][..., :7], "refined": refined[rows, selected],
                   "vels": cand_vels[rows, selected], "scores": fg_confidence[rows, selected].reshape(-1),
                   "labels": labels[rows], "ids": cur["ids"][rows]}
        return [self.update_trajectory(frame_id, pose, det_boxes, det_scores.clone(), det_vels, det_labels, matched)]

    def update_trajectory(self, frame_id, pose, det_boxes, det_scores, det_vels, det_labels, matched):
        """New tracks from confident detections that no surviving track explains (3-D IoU <= new_born_nms_thresh),
        then the bank / association update (:1286-1407)."""
        if frame_id > 0 and det_boxes.shape[0] > 0 and matched["boxes"].shape[0] > 0:
            explained = boxes_iou3d_gpu(det_boxes, matched["boxes"]).max(-1)[0] > self.new_born_nms_thresh
            det_scores = torch.where(explained, torch.zeros_like(det_scores), det_scores)
        gate = torch.full_like(det_scores, float("inf"))
        for c, t in self.new_born.items():
            gate = torch.where(det_labels == c, torch.full_like(gate, t), gate)
        born = (det_scores > gate).nonzero().flatten()
        dev = det_boxes.device
        if born.numel() > 0:
            new_ids = self.max_id + 1 + torch.arange(born.shape[0], device=dev)
            self.max_id = self.max_id + 1 + int(born.shape[0])
        else:
            new_ids = torch.zeros(0, dtype=torch.long, device=dev)
        ids = torch.cat([matched["ids"].to(dev).long(), new_ids])
        boxes = torch.cat([matched["boxes"], det_boxes[born]])
        refined = torch.cat([matched["refined"], det_boxes[born]])
        out, gbox, _ = self._store(ids, boxes, refined, torch.cat([matched["vels"], det_vels[born]]),
                                   torch.cat([matched["scores"], det_scores[born]]),
                                   torch.cat([matched["labels"], det_labels[born]]), pose)
        self.tracker.reset(gbox[:, :2], self.current["labels"].cpu().numpy())
        return out
