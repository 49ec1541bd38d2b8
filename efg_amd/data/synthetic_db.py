"""A small deterministic ground-truth object database + scenes for the DatabaseSampling parity tests (no dataset
access): 40 objects per class with 3-300 object-centred points (x y z intensity elongation) and a 7-column box placed
somewhere in the detection range; entries below the `min_points` filter and tight clusters (collisions, containment)
are included on purpose."""
import numpy as np

_SIZES = {"VEHICLE": (4.7, 2.1, 1.7), "PEDESTRIAN": (0.9, 0.9, 1.7), "CYCLIST": (1.8, 0.8, 1.7)}


def make_database(seed=7, per_class=40):
    rng = np.random.default_rng(seed)
    infos, clouds = {}, {}
    for name, size in _SIZES.items():
        infos[name] = []
        for k in range(per_class):
            n = int(rng.integers(3, 300))
            dims = np.array(size) * rng.uniform(0.85, 1.15)
            centre = np.array([*rng.uniform(-30, 30, 2), -1.8 + dims[2] / 2])
            if k % 7 == 0:                                   # clusters: candidates that collide with each other
                centre[:2] = rng.normal(0, 1.5, 2) + (12.0 if name == "VEHICLE" else -12.0)
            box = np.concatenate([centre, dims, [rng.uniform(-np.pi, np.pi)]]).astype(np.float32)
            if name == "PEDESTRIAN" and k % 9 == 1:         # a small box fully inside a database vehicle's footprint
                host = infos["VEHICLE"][k % len(infos["VEHICLE"])]["box3d_lidar"]
                box[:2] = host[:2]
            pts = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)) * dims, np.tanh(rng.uniform(0, 2, (n, 1))),
                                  rng.uniform(0, 1.5, (n, 1))], 1).astype(np.float32)
            path = "gt_database/%s_%03d.bin" % (name.lower(), k)
            infos[name].append({"name": name, "path": path, "box3d_lidar": box, "num_points_in_gt": n,
                                "difficulty": int(rng.integers(0, 3)), "gt_idx": k})
            clouds[path] = pts
    return infos, clouds


def make_sampling_scene(seed, n_points=4000):
    """(points [n, 5] float32, info with gt_boxes [k, 7], gt_names, difficulty, num_points_in_gt)."""
    rng = np.random.default_rng(seed)
    k = int(rng.integers(4, 12))
    names = rng.choice(list(_SIZES), k)
    boxes = np.zeros((k, 7), np.float32)
    for i, nm in enumerate(names):
        dims = np.array(_SIZES[nm]) * rng.uniform(0.9, 1.1)
        boxes[i] = [*rng.uniform(-28, 28, 2), -1.8 + dims[2] / 2, *dims, rng.uniform(-np.pi, np.pi)]
    pts = np.concatenate([rng.uniform(-35, 35, (n_points, 2)), rng.normal(-1.8, 0.05, (n_points, 1)),
                          np.tanh(rng.uniform(0, 2, (n_points, 1))), rng.uniform(0, 1.5, (n_points, 1))], 1).astype(np.float32)
    info = {"annotations": {"gt_boxes": boxes, "gt_names": np.array([str(n) for n in names]),
                            "difficulty": np.zeros(k, np.int64), "num_points_in_gt": np.full(k, 50, np.int64)}}
    return pts, info
