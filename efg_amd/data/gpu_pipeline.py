"""GPU-side point-cloud augmentation -- host mirror of the reference's train
processors ($CQ/config.yaml:32-42, efg/data/augmentations/extend_3d.py) over libefg_hip.so.

Same class names, constructor arguments and `__call__(points, info) -> (points, info)` protocol as the reference.
`points` is a `DevicePoints` (an [N, F] float32 cloud resident in HBM); `info["annotations"]` stays on the host as
numpy arrays, as in the reference.  Every processor draws its random numbers with the reference's numpy calls in
the reference's order, so a numpy seed produces the same augmentation as the reference pipeline; the per-point
arithmetic is deferred into an op list and executed by ONE kernel when the cloud is filtered or materialised
(efg_points_transform_filter_f32), instead of one numpy pass per processor on DataLoader workers.

`DatabaseSampling` (ground-truth paste) is in gt_database.py, with the object database resident in HBM; the CPU
`Voxelization` processor is not mirrored: the model voxelizes on the GPU (operators/voxelize.py).
"""
import ctypes

import numpy as np
import torch

from .. import _lib as L

NEG_Y, NEG_X, ROT_Z, SCALE, TRANSLATE = 0, 1, 2, 3, 4


class _PointOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("a", ctypes.c_float), ("b", ctypes.c_float), ("c", ctypes.c_float)]


class DevicePoints:
    """An [N, F] float32 cloud on the GPU plus the transforms queued on it."""

    def __init__(self, points, device=None):
        if isinstance(points, np.ndarray):  # host array: upload to `device` (default: the current GPU)
            if device is None:
                L.lib()  # no GPU library => the usual loud failure, before touching torch.cuda
                device = torch.device("cuda", torch.cuda.current_device())
            points = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(device)
        L.require_gpu(points)
        self.tensor = points.contiguous().float()
        self.ops = []

    def queue(self, kind, a=0.0, b=0.0, c=0.0):
        if len(self.ops) == 8:
            self.materialize()
        self.ops.append((kind, float(a), float(b), float(c)))

    def materialize(self, pc_range=None):
        """Apply the queued ops (and the range filter) in one pass; returns the number of rows (host int)."""
        t = self.tensor
        n, f = t.shape
        if not self.ops and pc_range is None:
            return n
        ops = (_PointOp * max(len(self.ops), 1))(*[_PointOp(*o) for o in self.ops])
        rng = L.host_f32(pc_range, 6) if pc_range is not None else None
        out = torch.empty_like(t)
        count = torch.empty(1, dtype=torch.int32, device=t.device)
        ws_bytes = L.lib().efg_points_transform_filter_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=t.device)
        L.check(L.lib().efg_points_transform_filter_f32(L.ptr(t), n, f, ops, len(self.ops), rng, L.ptr(out),
                                                        L.ptr(count), L.ptr(ws), ws_bytes, L.stream()))
        self.ops = []
        m = int(count.item()) if pc_range is not None else n   # the filter sizes the cloud: one read-back
        self.tensor = out[:m]
        return m

    def finalize(self):
        self.materialize()
        return self.tensor


def _rotate_z(xyz, angle):
    """rotate_points_along_z (box_ops.py:517-535) in fp32 on the host, for the (few) annotation boxes."""
    a = np.float32(angle)
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    x, y = xyz[:, 0].astype(np.float32), xyz[:, 1].astype(np.float32)
    out = xyz.astype(np.float32).copy()
    out[:, 0] = x * c + y * (-s)
    out[:, 1] = x * s + y * c
    return out


def _all_annotations(info):
    if "annotations" in info:
        yield info["annotations"]
        for sweep in info.get("sweeps", []):
            if "annotations" in sweep:
                yield sweep["annotations"]


class RandomFlip3D:
    """extend_3d.py:120-162."""

    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, points, info):
        if np.random.choice([False, True], replace=False, p=[1 - self.p, self.p]):   # flip along x axis
            points.queue(NEG_Y)
            for ann in _all_annotations(info):
                b = ann["gt_boxes"]
                b[:, 1] = -b[:, 1]
                b[:, -1] = -b[:, -1]
                if b.shape[1] > 7:
                    b[:, 7] = -b[:, 7]
        if np.random.choice([False, True], replace=False, p=[1 - self.p, self.p]):   # flip along y axis
            points.queue(NEG_X)
            for ann in _all_annotations(info):
                b = ann["gt_boxes"]
                b[:, 0] = -b[:, 0]
                b[:, -1] = -(b[:, -1] + np.pi)
                if b.shape[1] > 7:
                    b[:, 6] = -b[:, 6]
        return points, info


class GlobalRotation:
    """extend_3d.py:165-199."""

    def __init__(self, rotation):
        self.rotation = rotation if isinstance(rotation, list) else [-rotation, rotation]

    def __call__(self, points, info):
        noise_rotation = np.random.uniform(self.rotation[0], self.rotation[1])
        a = np.float32(noise_rotation)
        points.queue(ROT_Z, np.float32(np.cos(a)), np.float32(np.sin(a)))
        for ann in _all_annotations(info):
            b = ann["gt_boxes"]
            b[:, :3] = _rotate_z(b[:, :3], noise_rotation)
            b[:, -1] += noise_rotation
            if b.shape[1] > 7:
                vel = np.hstack([b[:, 6:8], np.zeros((b.shape[0], 1), b.dtype)])
                b[:, 6:8] = _rotate_z(vel, noise_rotation)[:, :2]
        return points, info


class GlobalScaling:
    """extend_3d.py:202-218."""

    def __init__(self, min_scale, max_scale):
        self.min_scale, self.max_scale = min_scale, max_scale

    def __call__(self, points, info):
        noise_scale = np.random.uniform(self.min_scale, self.max_scale)
        points.queue(SCALE, np.float32(noise_scale))
        for ann in _all_annotations(info):
            ann["gt_boxes"][:, :-1] *= noise_scale
        return points, info


class GlobalTranslation:
    """extend_3d.py:221-236."""

    def __init__(self, std=(0, 0, 0)):
        self.std = std

    def __call__(self, points, info):
        trans = np.random.normal(scale=np.array(self.std, dtype=np.float32), size=3).T
        points.queue(TRANSLATE, *[np.float32(v) for v in trans])
        for ann in _all_annotations(info):
            ann["gt_boxes"][:, :3] += trans
        return points, info


def _dict_select(d, keep):
    for k, v in d.items():
        if isinstance(v, dict):
            _dict_select(v, keep)
        else:
            d[k] = v[keep]


class FilterByRange:
    """extend_3d.py:286-315 with mask_points_by_range / mask_boxes_outside_range_bev_z_bound (box_ops.py:459-477,
    538-548)."""

    def __init__(self, pc_range, with_gt=True, with_data=True):
        self.pc_range = np.array(list(pc_range))
        self.with_gt, self.with_data = with_gt, with_data

    def _box_mask(self, boxes):
        r = self.pc_range
        m1 = (boxes[:, 0] >= r[0]) & (boxes[:, 0] <= r[3]) & (boxes[:, 1] >= r[1]) & (boxes[:, 1] <= r[4])
        zmax, zmin = boxes[:, 2] + boxes[:, 5] / 2, boxes[:, 2] - boxes[:, 5] / 2
        return m1 & ~((zmax < r[2]) ^ (zmin > r[5]))

    def __call__(self, points, info):
        if self.with_data:
            points.materialize(self.pc_range)
        if self.with_gt:
            for ann in _all_annotations(info):
                _dict_select(ann, self._box_mask(ann["gt_boxes"]))
        return points, info


class PointShuffle:
    """extend_3d.py:108-118: np.random.shuffle of the rows; the permutation is drawn on the host with the same
    RNG call (it only depends on the row count) and applied as a device gather."""

    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, points, info):
        if np.random.uniform(0, 1.0, []) <= self.p:
            m = points.materialize()
            perm = np.arange(m)
            np.random.shuffle(perm)
            t = points.tensor
            idx = torch.from_numpy(perm).to(t.device, non_blocking=True)
            out = torch.empty_like(t)
            L.check(L.lib().efg_points_gather_f32(L.ptr(t), L.ptr(idx), m, t.shape[1], L.ptr(out), L.stream()))
            points.tensor = out
        return points, info


def build_train_pipeline(pc_range, p_flip=0.5, rotation=0.78539816, min_scale=0.8, max_scale=1.2, p_shuffle=1.0,
                         database=None):
    """The ConQueR train chain of $CQ/config.yaml:24-42 (Voxelization happens inside the model, on the GPU);
    `database`: a `gt_database.DeviceGTDatabase` for the leading DatabaseSampling processor, or None to skip it."""
    chain = [RandomFlip3D(p_flip), GlobalRotation(rotation), GlobalScaling(min_scale, max_scale),
             FilterByRange(pc_range), PointShuffle(p_shuffle)]
    if database is not None:
        from .gt_database import DatabaseSampling

        chain.insert(0, DatabaseSampling(database))
    return chain


def run(pipeline, points, info):
    """points: [N, F] device tensor (or numpy, uploaded) -> (device tensor, info)."""
    dp = points if isinstance(points, DevicePoints) else DevicePoints(points)
    for proc in pipeline:
        dp, info = proc(dp, info)
    return dp.finalize(), info
