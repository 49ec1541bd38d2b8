"""CPU restatement of the reference's point-cloud augmentation chain -- TEST INFRASTRUCTURE ONLY (see oracle/__init__).

SURVEY.md section 8(f) row n3 ("GPU-side data pipeline"): the train processors of $CQ/config.yaml:32-42 --
RandomFlip3D, GlobalRotation, GlobalScaling, FilterByRange, PointShuffle
(efg/data/augmentations/extend_3d.py:108-118,120-162,165-199,202-218,286-315; helpers
efg/geometry/box_ops.py:459-477,517-548).  DatabaseSampling (GT paste from an on-disk database) is not part of it.

Each function takes the random draws as ARGUMENTS (`draw_params` makes them with the reference's numpy calls in
the reference's order), so the arithmetic can be compared independently of the RNG.  PINNED against the reference
classes imported in place (scripts/make_golden_augment.py -> tests/golden/augment_*.npz): bit-exact except the
rotation, which the reference evaluates with torch.matmul on fp32 (summation order / FMA use of the BLAS) -- the
restatement matches to 1 ulp-level (<= 2e-6 relative), and the filter decisions that depend on it are identical on
the golden inputs.
"""
import numpy as np


def draw_params(p_flip, rotation, min_scale, max_scale):
    """The random numbers of RandomFlip3D -> GlobalRotation -> GlobalScaling, drawn from the GLOBAL numpy RNG
    with the same calls in the same order as the reference (extend_3d.py:129,147,191,209)."""
    flip_x_axis = bool(np.random.choice([False, True], replace=False, p=[1 - p_flip, p_flip]))  # y -> -y
    flip_y_axis = bool(np.random.choice([False, True], replace=False, p=[1 - p_flip, p_flip]))  # x -> -x
    rot = rotation if isinstance(rotation, (list, tuple)) else [-rotation, rotation]
    angle = np.random.uniform(rot[0], rot[1])
    scale = np.random.uniform(min_scale, max_scale)
    return {"flip_x_axis": flip_x_axis, "flip_y_axis": flip_y_axis, "angle": angle, "scale": scale}


def shuffle_permutation(m, p_shuffle=1.0):
    """PointShuffle (extend_3d.py:108-118): `_rand_range() <= p` then np.random.shuffle on the rows.  The
    permutation np.random.shuffle applies depends only on the row count, so shuffling arange(m) with the same RNG
    state yields the row order of the shuffled cloud.  Returns None when the shuffle is skipped."""
    if np.random.uniform(0, 1.0, []) <= p_shuffle:
        perm = np.arange(m)
        np.random.shuffle(perm)
        return perm
    return None


def _rotate_z(xyz, angle):
    """rotate_points_along_z (box_ops.py:517-535): [x, y, z] @ [[c, s, 0], [-s, c, 0], [0, 0, 1]] in fp32."""
    a = np.float32(angle)
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    x, y = xyz[:, 0].astype(np.float32), xyz[:, 1].astype(np.float32)
    out = xyz.astype(np.float32).copy()
    out[:, 0] = x * c + y * (-s)
    out[:, 1] = x * s + y * c
    return out


def transform_points(points, prm):
    """Flip -> rotate -> scale on an [N, F] float32 cloud (F >= 3; extra features untouched)."""
    pts = np.array(points, dtype=np.float32, copy=True)
    if prm["flip_x_axis"]:
        pts[:, 1] = -pts[:, 1]
    if prm["flip_y_axis"]:
        pts[:, 0] = -pts[:, 0]
    pts[:, :3] = _rotate_z(pts[:, :3], prm["angle"])
    pts[:, :3] *= np.float32(prm["scale"])
    return pts


def transform_boxes(boxes, prm):
    """The same chain on [M, 7(+2)] boxes (x, y, z, dx, dy, dz, [vx, vy,] heading), extend_3d.py:131-156,171-184,
    211-213."""
    b = np.array(boxes, dtype=np.float32, copy=True)
    if prm["flip_x_axis"]:
        b[:, 1] = -b[:, 1]
        b[:, -1] = -b[:, -1]
        if b.shape[1] > 7:
            b[:, 7] = -b[:, 7]
    if prm["flip_y_axis"]:
        b[:, 0] = -b[:, 0]
        b[:, -1] = -(b[:, -1] + np.float32(np.pi))
        if b.shape[1] > 7:
            b[:, 6] = -b[:, 6]
    b[:, :3] = _rotate_z(b[:, :3], prm["angle"])
    b[:, -1] += np.float32(prm["angle"])
    if b.shape[1] > 7:
        vel = np.hstack([b[:, 6:8], np.zeros((b.shape[0], 1), np.float32)])
        b[:, 6:8] = _rotate_z(vel, prm["angle"])[:, :2]
    b[:, :-1] *= np.float32(prm["scale"])
    return b


def mask_points_by_range(points, pc_range):
    """box_ops.py:538-548 (bounds inclusive on both sides)."""
    r = [np.float32(v) for v in pc_range]
    p = points
    return ((p[:, 0] >= r[0]) & (p[:, 0] <= r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] <= r[4]) & (p[:, 2] >= r[2]) &
            (p[:, 2] <= r[5]))


def mask_boxes_bev_z_bound(boxes, pc_range):
    """mask_boxes_outside_range_bev_z_bound (box_ops.py:459-477): centre inside the BEV range and the box not
    entirely below / above the z range (the z extent of the 8 corners is z +- dz / 2)."""
    r = [np.float32(v) for v in pc_range]
    b = boxes
    m1 = (b[:, 0] >= r[0]) & (b[:, 0] <= r[3]) & (b[:, 1] >= r[1]) & (b[:, 1] <= r[4])
    zmax, zmin = b[:, 2] + b[:, 5] / 2, b[:, 2] - b[:, 5] / 2
    m2 = (zmax < r[2]) ^ (zmin > r[5])
    return m1 & ~m2


def pipeline(points, boxes, prm, pc_range, perm_fn=shuffle_permutation):
    """Full chain on one sample.  Returns (points, boxes, kept-box mask)."""
    pts = transform_points(points, prm)
    bxs = transform_boxes(boxes, prm)
    pts = pts[mask_points_by_range(pts, pc_range)]
    keep = mask_boxes_bev_z_bound(bxs, pc_range)
    perm = perm_fn(pts.shape[0])
    if perm is not None:
        pts = pts[perm]
    return pts, bxs[keep], keep
