"""Box / trajectory geometry of the TrajectoryFormer training path.

Counterpart of $TF/modules/utils.py ($TF = playground/tracking.3d/waymo/trajectoryformer/trajectoryformer.centerpoint):
same function names, argument meaning and results for the entry points the model calls, written batch-first for the
GPU: no per-ROI Python loop and no per-frame `.cuda()` allocations; the point crop synchronises only for its boolean
selections and for the occupancy counts of crowded ROIs (the reference sub-samples those with NumPy's generator).

Boxes are (x, y, z, dx, dy, dz, heading[, t]); trajectories are [batch, frame, track, hypothesis, 8].
"""
import functools

import numpy as np
import torch


def rotate_points_along_z(points, angle):
    """points [B, N, 3+C] (or [B, N, 2]), angle [B] -> rotated about +z, angle grows x -> y (utils.py:13-38)."""
    c, s = torch.cos(angle), torch.sin(angle)
    if points.shape[-1] == 2:
        rot = torch.stack((c, s, -s, c), dim=1).view(-1, 2, 2).float()
        return torch.matmul(points, rot)
    o, z = torch.ones_like(angle), torch.zeros_like(angle)
    rot = torch.stack((c, s, z, -s, c, z, z, z, o), dim=1).view(-1, 3, 3).float()
    xyz = torch.matmul(points[..., 0:3], rot)
    return torch.cat((xyz, points[..., 3:]), dim=-1)


def encode_boxes_res_torch(boxes, anchors):
    """Residual box code of `boxes` w.r.t. `anchors`, both (N, 7+C) (utils.py:41-73).  Unlike the reference this does
    not clamp its arguments in place; callers here pass temporaries, so nothing observable changes."""
    a_size = torch.clamp_min(anchors[:, 3:6], 1e-5)
    g_size = torch.clamp_min(boxes[:, 3:6], 1e-5)
    diagonal = torch.sqrt(a_size[:, 0:1] ** 2 + a_size[:, 1:2] ** 2)
    planar = (boxes[:, 0:2] - anchors[:, 0:2]) / diagonal
    zt = (boxes[:, 2:3] - anchors[:, 2:3]) / a_size[:, 2:3]
    size = torch.log(g_size / a_size)
    rest = boxes[:, 6:] - anchors[:, 6:]            # heading residual and any extra code
    return torch.cat([planar, zt, size, rest], dim=-1)


def decode_torch(box_encodings, anchors):
    """Inverse of `encode_boxes_res_torch` (utils.py:76-104 / losses.py:133-159)."""
    a_size = anchors[..., 3:6]
    diagonal = torch.sqrt(a_size[..., 0:1] ** 2 + a_size[..., 1:2] ** 2)
    planar = box_encodings[..., 0:2] * diagonal + anchors[..., 0:2]
    z = box_encodings[..., 2:3] * a_size[..., 2:3] + anchors[..., 2:3]
    size = torch.exp(box_encodings[..., 3:6]) * a_size
    rest = box_encodings[..., 6:] + anchors[..., 6:]
    return torch.cat([planar, z, size, rest], dim=-1)


_CORNER_TEMPLATE = ((1, 1, -1), (1, -1, -1), (-1, -1, -1), (-1, 1, -1), (1, 1, 1), (1, -1, 1), (-1, -1, 1), (-1, 1, 1))


@functools.lru_cache(maxsize=None)
def _constant(values, dtype, device):
    """A small constant tensor on `device`, built once: `tensor(list, device=gpu)` / `new_tensor(list)` are blocking uploads,
    i.e. a host wait for everything queued on the stream (found with scripts/ubench/sync_points.py)."""
    return torch.tensor(values, dtype=dtype, device=device)


def boxes_to_corners_3d(boxes3d):
    """(N, 7) -> (N, 8, 3) corners, bottom face first (utils.py:107-144)."""
    template = _constant(_CORNER_TEMPLATE, boxes3d.dtype, boxes3d.device) / 2
    corners = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None]
    corners = rotate_points_along_z(corners, boxes3d[:, 6])
    return corners + boxes3d[:, None, 0:3]


def _rotate_sequence(seq, angle):
    """seq [B, T, N, H, C] rotated per (b, n, h) by angle [B, 1, N, H] -- the permute/reshape round trip of
    utils.py:172-181 folded into one helper."""
    b, t, n, h, c = seq.shape
    flat = seq.permute(0, 2, 3, 1, 4).reshape(b * n * h, t, c)
    return rotate_points_along_z(flat, angle.reshape(-1)).reshape(b, n, h, t, c).permute(0, 3, 1, 2, 4)


def transform_trajs_to_local_coords(box_seq, center_xyz, center_heading, pred_vel_hypo=None, heading_index=8,
                                    rot_vel_index=(6, 7)):
    """Trajectories into the frame of their newest box (utils.py:147-199).  Entries whose anchor centre or own size is
    all-zero (padding) come back as zeros."""
    t = box_seq.shape[1]
    valid = torch.logical_and((center_xyz[..., :2].sum(-1) != 0).repeat(1, t, 1, 1), box_seq[..., 3:6].sum(-1) != 0)
    local = box_seq.clone()
    local[..., 0:2] = local[..., 0:2] - center_xyz[..., :2]
    local = _rotate_sequence(local, -center_heading).clone()
    local[..., heading_index] = local[..., heading_index] - center_heading
    keep = valid.unsqueeze(-1)
    vel = None
    if pred_vel_hypo is not None:
        vel = _rotate_sequence(pred_vel_hypo, -center_heading)
        vel = torch.where(keep, vel, torch.zeros_like(vel))
    return torch.where(keep, local, torch.zeros_like(local)), vel


def transform_trajs_to_global_coords(box_seq, center_xyz, center_heading, pred_vel_repeat=None, heading_index=6):
    """Inverse of the above (utils.py:202-240)."""
    out = _rotate_sequence(box_seq, center_heading).clone()
    d = center_xyz.shape[-1]
    out[..., 0:d] = out[..., 0:d] + center_xyz
    out[..., heading_index] = out[..., heading_index] + center_heading
    vel = None if pred_vel_repeat is None else _rotate_sequence(pred_vel_repeat, center_heading)
    return out, vel


def get_corner_points_of_roi(rois):
    """(..., 7+) -> (global [R, 8, 3], local [R, 8, 3]) corners in the order (z fastest) of `ones(2,2,2).nonzero()`
    (utils.py:301-325)."""
    rois = rois.reshape(-1, rois.shape[-1])
    size = rois[:, 3:6].unsqueeze(1)
    bits = _constant(tuple((i >> 2 & 1, i >> 1 & 1, i & 1) for i in range(8)), rois.dtype, rois.device)
    local = bits[None] * size - size / 2
    local = rotate_points_along_z(local, rois[:, 6])
    return local + rois[:, None, 0:3], local


def spherical_coordinate(src, diag_dist):
    """[.., 27] offsets to 8 corners + centre -> [.., 27] (range / diagonal, azimuth, inclination) blocks
    (utils.py:328-343)."""
    assert src.shape[-1] == 27
    x, y, z = src[..., 0::3], src[..., 1::3], src[..., 2::3]
    dis = (x ** 2 + y ** 2 + z ** 2) ** 0.5
    phi = torch.atan(y / (x + 1e-5))
    the = torch.acos(z / (dis + 1e-5))
    return torch.cat([dis / (diag_dist + 1e-5), phi, the], dim=-1)


def reorder_rois(pred_bboxes):
    """List of [n_i, C] -> zero-padded [len, max(n_i, 1), C] and its validity mask (utils.py:346-358)."""
    width = max(1, max(len(b) for b in pred_bboxes))
    first = pred_bboxes[0]
    out = first.new_zeros((len(pred_bboxes), width, first.shape[-1]), dtype=torch.float32)
    valid = torch.zeros(out.shape, dtype=torch.bool, device=out.device)
    for i, b in enumerate(pred_bboxes):
        out[i, :len(b)] = b
        valid[i, :len(b)] = True
    return out, valid


@functools.lru_cache(maxsize=8192)
def _crowded_draw(count, k):
    """(indices, generator state after the draw) of `np.random.seed(0); np.random.choice(count, k, replace=False)`."""
    keep = np.random.get_state()
    np.random.seed(0)
    picked = np.random.choice(count, k, replace=False).astype(np.int64)
    after = np.random.get_state()
    np.random.set_state(keep)
    return picked, after


def _crowded_choice(count, k):
    """The reference re-seeds NumPy with 0 before every sub-sampling draw (utils.py:409-413), so the selection is a
    pure function of the ROI's point count: memoised (a training step draws for ~250 crowded boxes; seeding the
    Mersenne twister and permuting `count` indices costs 30 us each).  The global generator is left exactly where the
    reference's draw leaves it, for whatever draws from it next (the hypothesis augmentation)."""
    picked, after = _crowded_draw(int(count), int(k))
    np.random.set_state(after)
    return torch.from_numpy(picked)


def _crowded_choices(counts, k):
    """`_crowded_choice` for a sequence of ROIs, in order -> int64 [len, k]; one generator update, for the last."""
    draws = [_crowded_draw(int(c), int(k)) for c in counts]
    np.random.set_state(draws[-1][1])
    return torch.from_numpy(np.stack([d[0] for d in draws]))


def _crop_select_gpu(first, clouds, k):
    """Membership of every (box, point) through csrc/crop.hip: per scene the boxes' cylinders against its points, all
    scenes in one launch.  Returns (cloud [sum P, 6], point_base [B], count [B, R] (host list), start [B, R] tensor,
    index [total] -- scene-local rows of the inside points, box by box in cloud order)."""
    from .. import _lib as L

    batch, n_rois = first.shape[:2]
    dev = first.device
    padded = (n_rois + 15) // 16 * 16
    sizes = [c.shape[0] for c in clouds]
    base = np.concatenate([[0], np.cumsum(sizes)])
    cloud = torch.cat([c.contiguous() for c in clouds], 0) if batch > 1 else clouds[0].contiguous()
    radius = torch.sqrt((first[..., 3] / 2) ** 2 + (first[..., 4] / 2) ** 2) * 1.2
    xyr = first.new_full((batch, padded, 3), -1.0)
    xyr[:, :n_rois, 0:2] = first[..., 0:2]
    xyr[:, :n_rois, 2] = radius
    rng = torch.as_tensor(np.repeat(np.stack([base[:-1], base[1:]], 1)[:, None, :], padded, 1), device=dev).contiguous()
    counts = torch.zeros(batch * padded, dtype=torch.int32, device=dev)
    groups = batch * padded // 16
    chunks = int(min(64, max(1, 2048 // max(groups, 1), 1)))     # ~2000 workgroups: 8 per CU
    per_chunk = torch.empty(batch * padded, chunks, dtype=torch.int32, device=dev)
    lib = L.lib()
    L.check(lib.efg_cylinder_select_f32(L.ptr(cloud), cloud.shape[0], cloud.shape[1], cloud.shape[1] - 1, 1.0, L.ptr(rng),
                                        L.ptr(xyr), batch * padded, None, L.ptr(counts), None, chunks, L.ptr(per_chunk),
                                        L.stream()))
    host = counts.view(batch, padded)[:, :n_rois].tolist()          # the step's one read-back for the crop
    flat_counts = counts.view(batch, padded).long()
    starts = (torch.cumsum(flat_counts.reshape(-1), 0) - flat_counts.reshape(-1)).contiguous()
    total = int(sum(sum(h) for h in host))
    index = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    if total:
        L.check(lib.efg_cylinder_select_f32(L.ptr(cloud), cloud.shape[0], cloud.shape[1], cloud.shape[1] - 1, 1.0,
                                            L.ptr(rng), L.ptr(xyr), batch * padded, L.ptr(starts), None, L.ptr(index),
                                            chunks, L.ptr(per_chunk), L.stream()))
    return cloud, base, host, flat_counts[:, :n_rois], starts.view(batch, padded)[:, :n_rois], index, total


def crop_current_frame_points(num_lidar_points, trajectory_rois, points):
    """Points of the current sweep inside the 1.2 x half-diagonal cylinder of every frame-0 hypothesis box.

    trajectory_rois [B, T, N, H, 8]; `points` either the reference's collated [P, 1+6] tensor (batch index first,
    utils.py:361-431) or a list of per-scene [P_b, 6] tensors.  Returns [B, N*H, num_lidar_points, 6]: the ROI's
    points in cloud order, padded with its first point; an empty ROI gets its own centre with zero features; a ROI
    with more than `num_lidar_points` points is sub-sampled with the reference's fixed NumPy draw.

    On the GPU the membership comes from `efg_cylinder_select_f32` (all scenes in two launches, no [boxes x points]
    matrix, one read-back); host tensors take the PyTorch formulation below (`_crop_scene_torch`).
    """
    batch, _, n_track, n_hypo, _ = trajectory_rois.shape
    n_rois = n_track * n_hypo
    first = trajectory_rois[:, 0].reshape(batch, n_rois, 8)
    k = num_lidar_points
    clouds = [points[b] if isinstance(points, (list, tuple)) else points[points[:, 0] == b][:, 1:] for b in range(batch)]
    centre_fill = torch.cat([first[:, :, None, :3].expand(-1, -1, k, -1), first.new_zeros(batch, n_rois, k, 3)], -1)
    if not first.is_cuda:
        return torch.stack([_crop_scene_torch(k, first[b], clouds[b], centre_fill[b]) for b in range(batch)])
    cloud, base, host, count, start, index, total = _crop_select_gpu(first, clouds, k)
    if total == 0:
        return centre_fill
    dev = first.device
    slot = torch.arange(k, device=dev)
    pick = torch.where(slot[None, None] < count[..., None], slot[None, None].expand(batch, n_rois, -1),
                       torch.zeros_like(slot)[None, None])
    crowded = [(b, r) for b in range(batch) for r, c in enumerate(host[b]) if c > k]
    if crowded:     # in (scene, box) order: leaves NumPy's generator where the reference's loop leaves it
        rows = _crowded_choices([host[b][r] for b, r in crowded], k).to(dev)
        flat = torch.as_tensor([b * n_rois + r for b, r in crowded], device=dev)
        pick = pick.reshape(batch * n_rois, k).index_copy(0, flat, rows).reshape(batch, n_rois, k)
    local = index[(start[..., None] + pick).clamp(max=total - 1)].long()
    rows = local + torch.as_tensor(base[:-1], device=dev)[:, None, None]
    gathered = cloud[rows.clamp(max=cloud.shape[0] - 1)]
    return torch.where((count == 0)[..., None, None], centre_fill, gathered)


def _crop_scene_torch(k, boxes8, cloud, centre_fill):
    """One scene with PyTorch ops (host tensors: the CPU tests and the golden comparison)."""
    n_rois = boxes8.shape[0]
    boxes = boxes8[:, :7]
    slot = torch.arange(k, device=boxes.device)
    radius = torch.sqrt((boxes[:, 3] / 2) ** 2 + (boxes[:, 4] / 2) ** 2) * 1.2
    if cloud.shape[0] == 0:
        return centre_fill
    dist = torch.norm(cloud[None, :, :2] - boxes[:, None, :2], dim=2)
    # (points of older sweeps, dt >= 1, are excluded by the membership test instead of a compaction of the cloud:
    # same points in the same order)
    inside = (dist <= radius[:, None]) & (cloud[:, -1] < 1)[None]
    count = inside.sum(1)
    count_host = count.tolist()
    total = sum(count_host)
    if total == 0:
        return centre_fill
    pairs = torch.nonzero_static(inside, size=total)   # (roi, point) in roi-major, cloud order
    start = torch.cumsum(count, 0) - count
    pick = torch.where(slot[None] < count[:, None], slot[None].expand(n_rois, -1), torch.zeros_like(slot)[None])
    crowded = [i for i, c in enumerate(count_host) if c > k]
    if crowded:
        rows = torch.stack([_crowded_choice(count_host[i], k) for i in crowded]).to(boxes.device)
        pick = pick.index_copy(0, torch.as_tensor(crowded, device=boxes.device), rows)
    flat = (start[:, None] + pick).clamp(max=total - 1)
    gathered = cloud[pairs[:, 1][flat]]
    return torch.where((count == 0)[:, None, None], centre_fill, gathered)
