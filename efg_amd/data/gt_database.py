"""Ground-truth paste augmentation with the object database resident in HBM.

Counterpart of `DatabaseSampling` (efg/data/augmentations/extend_3d.py:50-93) and its `DataBaseSampler` /
`BatchSampler` (efg/data/samplers/gt_database_sampler.py:16-211) -- the first processor of the ConQueR / CenterPoint
training pipelines ($CQ/config.yaml:24-31).  Same constructor arguments, `__call__(points, info)` protocol, NumPy
generator calls in the same order (a seed selects the same objects) and the same annotation updates.

What differs is where the bytes live: the reference opens one `.bin` file per pasted object per sample on DataLoader
workers; here every object cloud of the database is uploaded ONCE into a single [sum P, F] device buffer (a Waymo
GT database is a few GB; 288 GB of HBM) and a paste is one row gather + one translation + one concatenation on the
GPU.  Which objects are pasted is decided on the host exactly as in the reference: per class up to
`max - present` candidates from a shuffled cursor, rejected when their BEV box collides with a ground-truth box or
an earlier accepted candidate.  The collision test (`efg/geometry/box_ops.py:27-95`, a numba kernel there) is
vectorised NumPy with the compiled kernel's semantics (its `x is True` tests compare values under numba).
"""
import copy
import math

import numpy as np
import torch

from .gpu_pipeline import DevicePoints


def _rank_world():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class BatchSampler:
    """Shuffled cursor over a class's database entries, sharded by rank (gt_database_sampler.py:16-66)."""

    def __init__(self, sampled_list, name=None, shuffle=True):
        self.rank, self.num_replicas = _rank_world()
        self.num_samples = int(math.ceil(len(sampled_list) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas
        self._sampled_list = sampled_list
        self._shuffle = shuffle
        indices = np.arange(len(sampled_list)).tolist()
        if shuffle:
            np.random.shuffle(indices)
        indices += indices[:self.total_size - len(sampled_list)]
        self._indices = indices[self.num_samples * self.rank:self.num_samples * (self.rank + 1)]
        self._idx = 0
        self._name = name

    def sample(self, num):
        if self._idx + num >= self.num_samples:
            picked = self._indices[self._idx:].copy()
            if self._shuffle:
                np.random.shuffle(self._indices)
            self._idx = 0
        else:
            picked = self._indices[self._idx:self._idx + num]
            self._idx += num
        return [self._sampled_list[i] for i in picked]


def bev_corners(centers, dims, angles):
    """[n, 2], [n, 2], [n] -> [n, 4, 2] corners, clockwise from the minimum corner (box_ops.py:139-182, 561-577)."""
    unit = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], dims.dtype) - np.array(0.5, dims.dtype)
    corners = dims.reshape(-1, 1, 2) * unit.reshape(1, 4, 2)
    sin, cos = np.sin(angles), np.cos(angles)
    rot = np.stack([cos, sin, -sin, cos]).reshape(2, 2, -1)
    return np.einsum("aij,jka->aik", corners, rot) + centers.reshape(-1, 1, 2)


def box_collision_test(boxes, qboxes, clockwise=True):
    """[N, 4, 2] x [K, 4, 2] -> bool [N, K] (box_ops.py:27-95): axis-aligned hulls overlap AND (two edges cross OR
    one box lies strictly inside the other)."""
    lo, hi = boxes.min(1), boxes.max(1)
    qlo, qhi = qboxes.min(1), qboxes.max(1)
    iw = np.minimum(hi[:, None, 0], qhi[None, :, 0]) - np.maximum(lo[:, None, 0], qlo[None, :, 0])
    ih = np.minimum(hi[:, None, 1], qhi[None, :, 1]) - np.maximum(lo[:, None, 1], qlo[None, :, 1])
    near = (iw > 0) & (ih > 0)
    a, b = boxes[:, :, None, None, :], np.roll(boxes, -1, axis=1)[:, :, None, None, :]       # edges of boxes  [N,4,1,1,2]
    c, d = qboxes[None, None, :, :, :], np.roll(qboxes, -1, axis=1)[None, None, :, :, :]     # edges of qboxes [1,1,K,4,2]

    def ccw(p, q, r):
        return (r[..., 1] - p[..., 1]) * (q[..., 0] - p[..., 0]) > (q[..., 1] - p[..., 1]) * (r[..., 0] - p[..., 0])

    cross = (ccw(a, c, d) != ccw(b, c, d)) & (ccw(a, b, c) != ccw(a, b, d))                  # [N, 4, K, 4]
    crossing = cross.any(axis=(1, 3))

    def strictly_inside(outer, inner):
        """every corner of inner[j] strictly inside outer[i] -> [len(outer), len(inner)]"""
        vec = outer - np.roll(outer, -1, axis=1)
        if clockwise:
            vec = -vec
        diff = outer[:, None, :, None, :] - inner[None, :, None, :, :]                       # [O, I, edge k, corner L, 2]
        side = vec[:, None, :, None, 1] * diff[..., 0] - vec[:, None, :, None, 0] * diff[..., 1]
        return (side < 0).all(axis=(2, 3))

    contained = strictly_inside(boxes, qboxes) | strictly_inside(qboxes, boxes).T
    return near & (crossing | contained)


class DeviceGTDatabase:
    """The filtered database: per class its entries (box, name, difficulty, num_points_in_gt, row range in the
    device buffer) and a `BatchSampler`; all object clouds in one [sum P, F] float32 tensor on `device`."""

    def __init__(self, db_infos, clouds, groups, min_points=0, difficulty=-1, device=None, sample_func="sample"):
        """db_infos: {class name: [info dict with "box3d_lidar", "name", "difficulty", "num_points_in_gt", "path"]}
        (the reference's pickle); clouds: {info["path"]: float32 [P, F] object-centred points} or a callable
        path -> array (e.g. reading the reference's .bin files)."""
        self.min_points, self.difficulty = min_points, difficulty
        kept = {name: [i for i in infos if i["num_points_in_gt"] >= min_points and i["difficulty"] >= difficulty]
                for name, infos in db_infos.items()}
        self._sample_classes, self._sample_max_nums = [], []
        for group in groups:
            self._sample_classes += list(group.keys())
            self._sample_max_nums += list(group.values())
        rows, start = [], 0
        self.infos = {}
        for name, infos in kept.items():
            self.infos[name] = []
            for info in infos:
                cloud = clouds(info["path"]) if callable(clouds) else clouds[info["path"]]
                cloud = np.ascontiguousarray(cloud, np.float32)
                entry = dict(info)
                entry["rows"] = (start, start + cloud.shape[0])
                start += cloud.shape[0]
                rows.append(cloud)
                self.infos[name].append(entry)
        self.points = torch.from_numpy(np.concatenate(rows, 0) if rows else np.zeros((0, 5), np.float32))
        if device is not None:
            self.points = self.points.to(device)
        self._samplers = {name: BatchSampler(infos, name) for name, infos in self.infos.items()}
        self._sample_func = sample_func

    def _draw(self, name, num):
        if self._sample_func == "rand_sample":
            return list(copy.deepcopy(np.random.choice(self.infos[name], num)))
        return copy.deepcopy(self._samplers[name].sample(num))

    def sample_class(self, name, num, gt_boxes):
        """Candidates of one class that collide with nothing accepted so far (gt_database_sampler.py:187-211)."""
        sampled = self._draw(name, num)
        num_gt = gt_boxes.shape[0]
        sp_boxes = np.stack([i["box3d_lidar"] for i in sampled], axis=0)
        boxes = np.concatenate([gt_boxes, sp_boxes], axis=0).copy()
        corners = bev_corners(boxes[:, 0:2], boxes[:, 3:5], boxes[:, -1])
        coll = box_collision_test(corners, corners)
        np.fill_diagonal(coll, False)
        valid = []
        for i in range(num_gt, num_gt + len(sampled)):
            if coll[i].any():
                coll[i] = False
                coll[:, i] = False
            else:
                valid.append(sampled[i - num_gt])
        return valid

    def sample_all(self, gt_boxes, gt_names):
        """The accepted entries of every class, in class order (gt_database_sampler.py:112-176), or []."""
        accepted, avoid = [], gt_boxes
        for name, max_num in zip(self._sample_classes, self._sample_max_nums):
            num = int(np.round(int(max_num - np.sum([n == name for n in gt_names]))).astype(np.int64))
            if num > 0:
                got = self.sample_class(name, num, avoid)
                accepted += got
                if got:
                    avoid = np.concatenate([avoid, np.stack([s["box3d_lidar"] for s in got], axis=0)], axis=0)
        return accepted

    def gather(self, entries):
        """The pasted objects' points, moved to their boxes: one row gather + one add on the device."""
        idx = np.concatenate([np.arange(*e["rows"]) for e in entries])
        shift = np.concatenate([np.repeat(np.asarray(e["box3d_lidar"][:3], np.float32)[None], e["rows"][1] - e["rows"][0], 0)
                                for e in entries])
        dev = self.points.device
        pts = self.points.index_select(0, torch.from_numpy(idx).to(dev))
        pts[:, :3] += torch.from_numpy(shift).to(dev)
        return pts


def points_in_boxes(points, boxes):
    """[N, 3+] x [M, 7+] (x y z dx dy dz ... heading last) -> bool [N, M]; torch, any device."""
    rel = points[:, None, :3] - boxes[None, :, :3]
    c, s = torch.cos(boxes[:, -1]), torch.sin(boxes[:, -1])
    x = rel[..., 0] * c + rel[..., 1] * s
    y = -rel[..., 0] * s + rel[..., 1] * c
    half = boxes[None, :, 3:6] / 2
    return (x.abs() <= half[..., 0]) & (y.abs() <= half[..., 1]) & (rel[..., 2].abs() <= half[..., 2])


class DatabaseSampling:
    """`DatabaseSampling(db_info_path, sample_groups, min_points, difficulty, p, rm_points_after_sample)` of the
    reference, with `database` (a `DeviceGTDatabase`) in place of the path it would open."""

    def __init__(self, database, p=1.0, rm_points_after_sample=False):
        self.db_sampler = database
        self.p = p
        self.rm_points_after_sample = rm_points_after_sample

    def __call__(self, points, info):
        if np.random.uniform(0, 1.0, []) <= self.p:
            ann = info["annotations"]
            accepted = self.db_sampler.sample_all(ann["gt_boxes"], ann["gt_names"])
            if accepted:
                boxes = np.array([e["box3d_lidar"] for e in accepted])
                ann["gt_names"] = np.concatenate([ann["gt_names"], np.array([e["name"] for e in accepted])], axis=0)
                ann["gt_boxes"] = np.nan_to_num(np.concatenate([ann["gt_boxes"], boxes], axis=0))
                for key in ("difficulty", "num_points_in_gt"):
                    if key in ann:
                        ann[key] = np.concatenate([ann[key], np.array([e[key] for e in accepted])], axis=0)
                cloud = points.finalize() if isinstance(points, DevicePoints) else points
                pasted = self.db_sampler.gather(accepted).to(cloud.device)
                if self.rm_points_after_sample:
                    inside = points_in_boxes(cloud, torch.from_numpy(np.nan_to_num(boxes)).to(cloud))
                    cloud = cloud[~inside.any(-1)]
                merged = torch.nan_to_num(torch.cat([pasted, cloud], 0))
                if isinstance(points, DevicePoints):
                    points.tensor = merged
                else:
                    points = merged
        return points, info
