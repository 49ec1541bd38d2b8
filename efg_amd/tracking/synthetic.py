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


def make_tracking_sequence(seed, frames=8, n_objects=10, n_ground=4000, per_object=160):
    """A short drive for the online tracker ($TF/trajectoryformer.py:forward_inference): objects with constant global
    velocity, an ego pose that advances and yaws a little every 0.1 s, and per frame the detector's boxes in the
    vehicle frame (jitter, misses, a few low-score ghosts), a small current-sweep cloud (ground + object surfaces,
    columns x y z intensity elongation dt) and the reference's frame token.  Returns a list of `(sample, info)`."""
    rng = np.random.default_rng(seed)
    labels = rng.choice([1, 1, 1, 2, 3], n_objects).astype(np.float32)
    size = np.array([{1: (4.6, 2.0, 1.7), 2: (0.9, 0.8, 1.8), 3: (1.8, 0.7, 1.7)}[int(c)] for c in labels])
    size = size * rng.uniform(0.9, 1.1, (n_objects, 1))
    pos0 = np.concatenate([rng.uniform(-35, 35, (n_objects, 2)), -1.8 + size[:, 2:3] / 2], 1)
    vel = rng.normal(0, 3.0, (n_objects, 2)) * (labels[:, None] == 1) + rng.normal(0, 0.8, (n_objects, 2))
    yaw = np.arctan2(vel[:, 1], vel[:, 0])
    out = []
    for f in range(frames):
        t = 0.1 * f
        ego_yaw, ego_xy = 0.02 * f, np.array([6.0 * t, 0.4 * t])
        c, s = np.cos(ego_yaw), np.sin(ego_yaw)
        pose = np.array([[c, -s, 0, ego_xy[0]], [s, c, 0, ego_xy[1]], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
        rot = pose[:2, :2]
        centre = (pos0[:, :2] + vel * t - ego_xy) @ rot                       # R^T (p - t) for row vectors
        det = np.zeros((n_objects, 9))
        det[:, 0:2] = centre + rng.normal(0, 0.06, (n_objects, 2))
        det[:, 2] = pos0[:, 2] + rng.normal(0, 0.03, n_objects)
        det[:, 3:6] = size * (1 + rng.normal(0, 0.02, (n_objects, 3)))
        det[:, 6:8] = vel @ rot + rng.normal(0, 0.2, (n_objects, 2))
        det[:, 8] = yaw - ego_yaw + rng.normal(0, 0.03, n_objects)
        score = rng.uniform(0.55, 0.97, n_objects)
        seen = rng.uniform(size=n_objects) > 0.15
        n_ghost = 3
        ghost = np.zeros((n_ghost, 9))
        ghost[:, 0:2] = rng.uniform(-35, 35, (n_ghost, 2))
        ghost[:, 2] = -0.9
        ghost[:, 3:6] = rng.uniform(0.6, 4.0, (n_ghost, 3))
        ghost[:, 8] = rng.uniform(-np.pi, np.pi, n_ghost)
        boxes = np.concatenate([det[seen], ghost]).astype(np.float32)
        scores = np.concatenate([score[seen], rng.uniform(0.05, 0.9, n_ghost)]).astype(np.float32)
        lab = np.concatenate([labels[seen], rng.integers(1, 4, n_ghost).astype(np.float32)]).astype(np.float32)
        ground = np.concatenate([rng.uniform(-40, 40, (n_ground, 2)), rng.normal(-1.8, 0.03, (n_ground, 1))], 1)
        surf = []
        for k in range(n_objects):
            u = rng.uniform(-0.5, 0.5, (per_object, 3)) * size[k]
            cy, sy = np.cos(det[k, 8]), np.sin(det[k, 8])
            surf.append(np.stack([centre[k, 0] + cy * u[:, 0] - sy * u[:, 1], centre[k, 1] + sy * u[:, 0] + cy * u[:, 1],
                                  pos0[k, 2] + u[:, 2]], 1))
        xyz = np.concatenate([ground] + surf)
        pts = np.concatenate([xyz, np.tanh(rng.uniform(0, 2, (len(xyz), 1))), rng.uniform(0, 1.5, (len(xyz), 1)),
                              np.zeros((len(xyz), 1))], 1).astype(np.float32)
        pts = pts[rng.permutation(len(pts))]
        info = {"token": "segment-%d_frame_%d.npy" % (seed, f), "veh_to_global": pose,
                "annotations": {"pred_boxes3d": boxes, "pred_scores": scores, "pred_labels": lab}}
        out.append(([{"points": np.ascontiguousarray(pts)}], info))
    return out
