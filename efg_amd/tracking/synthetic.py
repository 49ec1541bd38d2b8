"""Synthetic TrajectoryFormer samples in the reference's input format ($TF/env.py:109-355): five-sweep points
(x, y, z, intensity, elongation, dt) plus detector boxes of eleven frames -- the current detections, the previous
frame's detections moved to the current time by their velocity, and the detections of the nine frames before --
each frame zero-padded to the widest one and flattened to [(traj_length + 1) * M, 9] / [..] scores / [..] labels.

There is no network for the Waymo CenterPoint box dumps; these are ground-truth boxes of a synthetic scene
(efg_amd/data/synthetic.py) jittered like detector output, with misses and low-score false positives.
"""
import numpy as np

from ..data.synthetic import make_scene


def make_tracking_sample(seed, n_points=150000, n_objects=40, traj_length=10, n_false=12, max_roi_num=128,
                         pc_range=None):
    kw = {} if pc_range is None else {"pc_range": pc_range}
    points, gt_boxes, labels = make_scene(seed, n_points=n_points, n_sweeps=5, n_boxes=n_objects, **kw)
    rng = np.random.default_rng(seed + 7919)
    reach = float(np.abs(gt_boxes[:, :2]).max()) if len(gt_boxes) else 10.0
    frames = []
    for f in range(traj_length + 1):
        age = 0 if f <= 1 else f - 1                     # frames 0 and 1 sit at the current time
        det = gt_boxes.astype(np.float64).copy()
        det[:, 0:2] -= 0.1 * age * det[:, 6:8]
        det[:, 0:3] += rng.normal(0, 0.08, (len(det), 3))
        det[:, 3:6] *= 1.0 + rng.normal(0, 0.03, (len(det), 3))
        det[:, 6:8] += rng.normal(0, 0.15, (len(det), 2))
        det[:, 8] += rng.normal(0, 0.04, len(det))
        seen = rng.uniform(size=len(det)) > 0.1
        score = rng.uniform(0.35, 0.95, len(det))
        ghost = np.zeros((n_false, 9))
        ghost[:, 0:2] = rng.uniform(-reach, reach, (n_false, 2))
        ghost[:, 2] = rng.normal(-0.9, 0.2, n_false)
        ghost[:, 3:6] = rng.uniform(0.5, 4.5, (n_false, 3))
        ghost[:, 8] = rng.uniform(-np.pi, np.pi, n_false)
        rows = np.concatenate([
            np.concatenate([det[seen], score[seen, None], labels[seen, None].astype(np.float64)], 1),
            np.concatenate([ghost, rng.uniform(0.02, 0.3, (n_false, 1)),
                            rng.integers(1, 4, (n_false, 1)).astype(np.float64)], 1)], 0)
        rows = rows[rng.permutation(len(rows))][:max_roi_num]
        frames.append(rows)
    width = max(1, max(len(r) for r in frames))
    table = np.zeros((traj_length + 1, width, 11), np.float32)
    for f, rows in enumerate(frames):
        table[f, :len(rows)] = rows
    table = table.reshape(-1, 11)
    annotations = {"gt_boxes": gt_boxes, "labels": labels, "difficulty": np.zeros(len(labels), np.int64),
                   "num_points_in_gt": np.full(len(labels), 50, np.int64),
                   "pred_boxes3d": np.ascontiguousarray(table[:, :9]), "pred_scores": np.ascontiguousarray(table[:, 9]),
                   "pred_labels": np.ascontiguousarray(table[:, 10])}
    return [{"points": points}], {"annotations": annotations}


def synthetic_tracking_batch(seed_base, scenes, device=None, **kw):
    """`scenes` samples; with `device` the points are uploaded once (the boxes stay NumPy, as the loader hands them)."""
    import torch

    batch = []
    for i in range(scenes):
        sample, info = make_tracking_sample(seed_base + i, **kw)
        if device is not None:
            sample[0]["points"] = torch.from_numpy(sample[0]["points"]).to(device)
            if sample[0]["points"].is_cuda:  # "the upload is complete once this fires" (see engine.synthetic_batch)
                sample[0]["ready_event"] = torch.cuda.Event()
                sample[0]["ready_event"].record(torch.cuda.current_stream(sample[0]["points"].device))
        batch.append((sample, info))
    return batch
