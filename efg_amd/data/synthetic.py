"""Deterministic "Waymo-shaped" synthetic scenes.

There is no dataset access (and the reference's loaders, efg/data/datasets/waymo/waymo.py:33-140,
are out of scope): benchmarks and parity tests use clouds generated here.  A scene is a 64-beam
spinning-lidar sweep over a ground plane, a few dozen planar facades and ``n_boxes`` object boxes
(vehicle / pedestrian / cyclist sized, Waymo label ids 1-3), with the reference's feature format:
``XYZIT``-style rows ``(x, y, z, intensity=tanh(.), elongation)`` for one sweep
(efg/data/datasets/waymo/utils.py:76) and an extra time channel for multi-sweep clouds
(playground/.../centerpoint...4f.improved/config.yaml:64).  Points are shuffled
(``PointShuffle p=1.0``, ConQueR config.yaml:41-42) and cropped to ``pc_range``
(``FilterByRange``).

seed convention: ``1000 * config_id + scene_idx``.
"""
import numpy as np

PC_RANGE = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
VOXEL_SIZE = (0.1, 0.1, 0.15)

_BOX_SIZES = {1: (4.7, 2.1, 1.7), 2: (0.9, 0.9, 1.7), 3: (1.8, 0.8, 1.7)}


def _ray_box_hits(origin_dirs, boxes):
    """Slab test of unit rays (from the origin) against yawed boxes; returns (t_hit[n], hit_box[n])."""
    n = origin_dirs.shape[0]
    t_best = np.full(n, np.inf, np.float64)
    for bx in boxes:
        cx, cy, cz, l, w, h, yaw = bx[:7]
        c, s = np.cos(-yaw), np.sin(-yaw)
        # rotate ray dirs / origin into the box frame
        dx = origin_dirs[:, 0] * c - origin_dirs[:, 1] * s
        dy = origin_dirs[:, 0] * s + origin_dirs[:, 1] * c
        dz = origin_dirs[:, 2]
        ox = -(cx * c - cy * s)
        oy = -(cx * s + cy * c)
        oz = -cz
        tmin = np.full(n, -np.inf)
        tmax = np.full(n, np.inf)
        for o, d, half in ((ox, dx, l / 2), (oy, dy, w / 2), (oz, dz, h / 2)):
            with np.errstate(divide="ignore", invalid="ignore"):
                t1 = (-half - o) / d
                t2 = (half - o) / d
            lo, hi = np.minimum(t1, t2), np.maximum(t1, t2)
            tmin = np.maximum(tmin, np.nan_to_num(lo, nan=-np.inf))
            tmax = np.minimum(tmax, np.nan_to_num(hi, nan=np.inf))
        hit = (tmax >= tmin) & (tmin > 0.5)
        t_best = np.where(hit & (tmin < t_best), tmin, t_best)
    return t_best


def make_scene(seed, n_points=180000, n_sweeps=1, n_boxes=40, pc_range=PC_RANGE, clutter=0.0):
    """Returns (points[N, 5 or 6] float32, gt_boxes[n_boxes, 9] float32, labels[n_boxes] int64).

    clutter: share of the returns replaced by isolated "vegetation" returns (uniform in range and height), each in a
    voxel of its own -- 0.55 gives > 120 000 occupied 0.1 m voxels per 180k-point sweep, so that the voxelizer's
    `max_voxels` cap and its `break` semantics are exercised at benchmark size (`bench.py --dense`).

    gt_boxes columns: x, y, z, l, w, h, vx, vy, yaw (Waymo info layout consumed by
    VoxelBoxCoder3D._encode, playground/.../modules/box_coder.py:50-70: columns [0..5] and [-1]).
    """
    rng = np.random.default_rng(seed)
    # ---- objects --------------------------------------------------------------------------
    labels = rng.choice([1, 2, 3], size=n_boxes, p=[0.6, 0.25, 0.15]).astype(np.int64)
    r = rng.uniform(6.0, 62.0, n_boxes)
    th = rng.uniform(-np.pi, np.pi, n_boxes)
    boxes = np.zeros((n_boxes, 9), np.float64)
    boxes[:, 0] = r * np.cos(th)
    boxes[:, 1] = r * np.sin(th)
    for i, lb in enumerate(labels):
        l, w, h = _BOX_SIZES[int(lb)]
        sc = rng.uniform(0.9, 1.1)
        boxes[i, 3:6] = (l * sc, w * sc, h * sc)
    boxes[:, 2] = -1.8 + boxes[:, 5] / 2
    boxes[:, 6:8] = rng.normal(0, 1.0, (n_boxes, 2))
    boxes[:, 8] = rng.uniform(-np.pi, np.pi, n_boxes)
    # ---- facades: vertical planar patches (buildings / walls) ------------------------------
    n_fac = 36
    fr = rng.uniform(12.0, 70.0, n_fac)
    fth = rng.uniform(-np.pi, np.pi, n_fac)
    facades = np.zeros((n_fac, 7))
    facades[:, 0] = fr * np.cos(fth)
    facades[:, 1] = fr * np.sin(fth)
    facades[:, 3] = rng.uniform(8.0, 30.0, n_fac)   # length
    facades[:, 4] = 0.3                              # thickness
    facades[:, 5] = rng.uniform(3.0, 6.0, n_fac)     # height
    facades[:, 2] = -1.8 + facades[:, 5] / 2
    facades[:, 6] = fth + np.pi / 2 + rng.normal(0, 0.4, n_fac)
    solids = np.concatenate([boxes[:, [0, 1, 2, 3, 4, 5, 8]], facades], 0)

    sweeps = []
    per_sweep = int(np.ceil(n_points / n_sweeps))
    for sw in range(n_sweeps):
        # ---- rays: 64 beams x 2800 azimuth steps + 10k near-range rays ------------------
        elev = np.deg2rad(np.linspace(2.4, -17.6, 64))
        azim = np.linspace(-np.pi, np.pi, 2800, endpoint=False) + rng.uniform(0, 2 * np.pi / 2800)
        ee, aa = np.meshgrid(elev, azim, indexing="ij")
        ee, aa = ee.ravel(), aa.ravel()
        ne = rng.uniform(np.deg2rad(-35.0), np.deg2rad(-12.0), 10000)
        na = rng.uniform(-np.pi, np.pi, 10000)
        ee, aa = np.concatenate([ee, ne]), np.concatenate([aa, na])
        dirs = np.stack([np.cos(ee) * np.cos(aa), np.cos(ee) * np.sin(aa), np.sin(ee)], -1)
        # ego motion between sweeps: objects / facades shift a little
        shift = np.array([0.35 * sw, 0.02 * sw, 0.0])
        sol = solids.copy()
        sol[:, :3] -= shift
        t_obj = _ray_box_hits(dirs, sol)
        with np.errstate(divide="ignore"):
            t_gnd = np.where(dirs[:, 2] < -1e-3, -1.8 / dirs[:, 2], np.inf)
        t = np.minimum(t_obj, t_gnd)
        keep = np.isfinite(t) & (t < 75.0 * 1.45)
        t, d = t[keep], dirs[keep]
        pts = d * t[:, None] + rng.normal(0, 0.02, (t.shape[0], 3))
        inten = np.tanh(rng.uniform(0, 2, t.shape[0]))
        elong = rng.uniform(0, 1.5, t.shape[0])
        if clutter > 0:
            veg = rng.uniform(size=t.shape[0]) < clutter
            nv = int(veg.sum())
            rr = rng.uniform(4.0, 74.0, nv)
            aa2 = rng.uniform(-np.pi, np.pi, nv)
            pts[veg] = np.stack([rr * np.cos(aa2), rr * np.sin(aa2), rng.uniform(-1.7, 3.5, nv)], 1)
        cols = [pts, inten[:, None], elong[:, None]]
        if n_sweeps > 1:
            cols.append(np.full((t.shape[0], 1), 0.1 * sw))
        p = np.concatenate(cols, 1)
        m = ((p[:, 0] >= pc_range[0]) & (p[:, 0] < pc_range[3]) & (p[:, 1] >= pc_range[1]) & (p[:, 1] < pc_range[4])
             & (p[:, 2] >= pc_range[2]) & (p[:, 2] < pc_range[5]))
        p = p[m]
        if p.shape[0] > per_sweep:
            p = p[rng.choice(p.shape[0], per_sweep, replace=False)]
        sweeps.append(p)
    points = np.concatenate(sweeps, 0)
    if points.shape[0] > n_points:
        points = points[rng.choice(points.shape[0], n_points, replace=False)]
    points = points[rng.permutation(points.shape[0])]
    return points.astype(np.float32), boxes.astype(np.float32), labels
