"""Shared by scripts/make_golden_full.py (reference side, build container) and tests/test_model_full_golden.py (our
side, CPU + GPU box): the reduced configuration, the deterministic weights and the deterministic inputs of the
full-model golden.  Weights are not stored in the fixture (17 M parameters); both sides fill the state dict --
whose names and shapes are identical by construction of the drop-in -- from the same name-keyed CPU generator.
"""
import zlib

import numpy as np
import torch

# pc_range +-6.4 m => grid 128 x 128 x 40, p3 = 16 x 16 tokens; real channel plan
FULL_OVERRIDES = {
    "dataset.pc_range": [-6.4, -6.4, -2.0, 6.4, 6.4, 4.0],
    "model.transformer.num_queries": 30,
    "model.transformer.enc_layers": 1,
}


# inference fixtures: 120 queries x 3 classes = 360 candidates, so that Voxel-DETR's fixed top-300 rule selects
INFER_OVERRIDES = {"model.transformer.num_queries": 120}


def deterministic_state(state):
    """name/shape -> value.  Scales are chosen so that activations stay O(1) through 21 sparse convs + the DETR."""
    out = {}
    for name in sorted(state):
        ref = state[name]
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
        if not torch.is_floating_point(ref):  # num_batches_tracked
            out[name] = ref.clone()
            continue
        shape = tuple(ref.shape)
        if name.startswith("transformer.decoder_gt."):
            continue  # filled below: a copy of the decoder, as at construction (voxel_detr.py:86-89)
        if ref.dim() >= 2:
            fan_in = max(ref.numel() // shape[0], 1)
            v = torch.randn(shape, generator=g) * (1.0 / fan_in ** 0.5)
            if "linear_box_weight" in name or "linear_attn_weight" in name:
                v = torch.randn(shape, generator=g) * 0.05  # zero-initialised in the reference: make them matter
        elif name.endswith("running_var"):
            v = 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            v = 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("linear_box_bias"):
            v = torch.rand(shape, generator=g)
        elif name.endswith(".weight") and ref.dim() == 1:
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)  # norm scales
        else:
            v = 0.02 * torch.randn(shape, generator=g)
        out[name] = v.to(ref.dtype)
    for name in state:
        if name.startswith("transformer.decoder_gt."):
            out[name] = out[name.replace("transformer.decoder_gt.", "transformer.decoder.", 1)].clone()
    return out


def full_inputs(seed=77, n_points=1500):
    """Two small scenes inside the +-6.4 m range: ground returns, a few walls and boxes with surface points;
    annotations in the Waymo layout (gt_boxes [k, 9] = x y z l w h vx vy yaw, labels 1..3)."""
    rng = np.random.default_rng(seed)
    points_list, annos = [], []
    for scene, n_boxes in enumerate((5, 3)):
        n_ground = n_points // 2
        xy = rng.uniform(-6.3, 6.3, (n_ground, 2))
        ground = np.concatenate([xy, rng.normal(-1.7, 0.03, (n_ground, 1))], 1)
        n_wall = n_points // 5
        wall_x = rng.uniform(-6.0, 6.0, n_wall)
        wall = np.stack([wall_x, np.full(n_wall, 4.0 - scene) + rng.normal(0, 0.02, n_wall), rng.uniform(-1.7, 2.5, n_wall)], 1)
        boxes, labels, obj = [], [], []
        per = (n_points - n_ground - n_wall) // n_boxes
        for b in range(n_boxes):
            lab = int(rng.integers(1, 4))
            size = {1: (4.6, 2.0, 1.7), 2: (0.9, 0.8, 1.8), 3: (1.8, 0.7, 1.7)}[lab]
            cx, cy = rng.uniform(-4.5, 4.5, 2)
            yaw = rng.uniform(-np.pi, np.pi)
            cz = -1.7 + size[2] / 2
            boxes.append([cx, cy, cz, size[0], size[1], size[2], 0.0, 0.0, yaw])
            labels.append(lab)
            u = rng.uniform(-0.5, 0.5, (per, 3)) * np.array(size)
            face = rng.integers(0, 3, per)
            sign = rng.choice([-0.5, 0.5], per)
            u[np.arange(per), face] = sign * np.array(size)[face]
            c, s = np.cos(yaw), np.sin(yaw)
            obj.append(np.stack([cx + c * u[:, 0] - s * u[:, 1], cy + s * u[:, 0] + c * u[:, 1], cz + u[:, 2]], 1))
        xyz = np.concatenate([ground, wall] + obj, 0)
        xyz = xyz[(np.abs(xyz[:, 0]) < 6.39) & (np.abs(xyz[:, 1]) < 6.39) & (xyz[:, 2] > -1.99) & (xyz[:, 2] < 3.99)]
        feats = np.stack([np.tanh(rng.uniform(0, 2, len(xyz))), rng.uniform(0, 1.5, len(xyz))], 1)
        pts = np.concatenate([xyz, feats], 1).astype(np.float32)
        pts = pts[rng.permutation(len(pts))]
        points_list.append(np.ascontiguousarray(pts))
        k = len(labels)
        annos.append({"gt_boxes": np.array(boxes, np.float32), "labels": np.array(labels, np.int64),
                      "difficulty": np.zeros(k, np.int64), "num_points_in_gt": np.full(k, 50, np.int64)})
    return points_list, annos

# CenterPoint at the same reduced range: grid 128 x 128 x 40, BEV 16 x 16 after the 8x reduction of the backbone
CENTERPOINT_OVERRIDES = {
    "dataset.pc_range": [-6.4, -6.4, -2.0, 6.4, 6.4, 4.0],
    "model.loss.max_objs": 20,
    "model.post_process.post_center_limit_range": [-8.0, -8.0, -10.0, 8.0, 8.0, 10.0],
    # random-weight boxes are ~1 m wide on a 0.8 m lattice (IoU ~0.1 between neighbours): a low threshold and a tight
    # post-NMS cap make the inference fixture exercise suppression and truncation
    "model.post_process.nms.nms_iou_threshold": 0.08,
    "model.post_process.nms.nms_post_max_size": 60,
}

# TrajectoryFormer: the reference configuration as is (hidden 256, 3 + 3 encoder layers); small scenes
TRACKING_SAMPLE = {"n_points": 40000, "n_objects": 10, "n_false": 4}


def tracking_inputs(seed=500, scenes=2):
    from efg_amd.tracking.synthetic import make_tracking_sample

    return [make_tracking_sample(seed + i, **TRACKING_SAMPLE) for i in range(scenes)]

# the online tracker: 8 frames of a 10-object drive
ONLINE_SEQUENCE = {"seed": 321, "frames": 8, "n_objects": 10}
