"""CenterPoint label assignment ($CP1/voxelnet.py:43-187, $CP1/center_utils.py:10-63): Gaussian heat-map peaks and
box codes at the BEV cell of every ground-truth centre.  Host code, as in the reference (annotations are host arrays; a
scene has a few dozen objects); the arrays are uploaded once per batch by the model."""
import numpy as np


def gaussian_radius(det_size, min_overlap=0.5):
    """Largest radius such that a box shifted by it still overlaps the ground truth by `min_overlap` (CornerNet)."""
    height, width = det_size
    b1 = height + width
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * width * height * (1 - min_overlap) / (1 + min_overlap))) / 2
    b2 = 2 * (height + width)
    r2 = (b2 + np.sqrt(b2 ** 2 - 16 * (1 - min_overlap) * width * height)) / 2
    a3, b3 = 4 * min_overlap, -2 * min_overlap * (height + width)
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * (min_overlap - 1) * width * height)) / 2
    return min(r1, r2, r3)


def _gaussian_patch(radius):
    d = 2 * radius + 1
    sigma = d / 6
    y, x = np.ogrid[-radius:radius + 1, -radius:radius + 1]
    g = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    return g


def draw_gaussian(heatmap, center, radius):
    """heatmap[y, x] = max(heatmap, gaussian centred at int(center)) clipped to the map."""
    g = _gaussian_patch(radius)
    x, y = int(center[0]), int(center[1])
    h, w = heatmap.shape
    left, right = min(x, radius), min(w - x, radius + 1)
    top, bottom = min(y, radius), min(h - y, radius + 1)
    if right > -left and bottom > -top:
        view = heatmap[y - top:y + bottom, x - left:x + right]
        patch = g[radius - top:radius + bottom, radius - left:radius + right]
        if min(view.shape) > 0 and min(patch.shape) > 0:
            np.maximum(view, patch, out=view)
    return heatmap


def _limit_period(val, offset=0.5, period=2 * np.pi):
    return val - np.floor(val / period + offset) * period


def assign_scene(annotations, tasks, class_names_plain, grid_size, pc_range, voxel_size, out_size_factor,
                 gaussian_overlap, max_objs, min_radius):
    """One scene -> {"hm": [T x (C_t, fy, fx)], "anno_box": [T x (max_objs, 10)], "ind", "mask", "cat": [T x (max_objs,)],
    "gt_boxes_and_cls": (max_objs, 10)}.  annotations: {"gt_boxes": [n, 9] (x y z l w h vx vy yaw), "gt_names": [n]}
    (or integer "labels" 1..K in the order of `class_names_plain`)."""
    boxes = np.asarray(annotations["gt_boxes"], dtype=np.float32)
    if "gt_names" in annotations:
        names = np.asarray(annotations["gt_names"])
        known = np.array([n in class_names_plain for n in names], dtype=bool)
        boxes, names = boxes[known], names[known]
        classes = np.array([class_names_plain.index(n) + 1 for n in names], dtype=np.int32)
    else:
        classes = np.asarray(annotations["labels"], dtype=np.int32)
        known = (classes >= 1) & (classes <= len(class_names_plain))
        boxes, classes = boxes[known], classes[known]
    fmap = np.asarray(grid_size[:2]) // out_size_factor  # (fx, fy)
    out = {"hm": [], "anno_box": [], "ind": [], "mask": [], "cat": []}
    first = 0
    all_boxes, all_classes = [], []
    for task in tasks:
        n_cls = len(task["class_names"])
        # objects of the task, grouped by class (the reference concatenates per-class selections)
        sel = np.concatenate([np.nonzero(classes == first + c + 1)[0] for c in range(n_cls)]) if n_cls else np.zeros(0, int)
        tb = boxes[sel].copy()
        tc = classes[sel] - first
        tb[:, -1] = _limit_period(tb[:, -1])
        hm = np.zeros((n_cls, int(fmap[1]), int(fmap[0])), dtype=np.float32)
        anno = np.zeros((max_objs, 10), dtype=np.float32)
        ind = np.zeros(max_objs, dtype=np.int64)
        mask = np.zeros(max_objs, dtype=np.uint8)
        cat = np.zeros(max_objs, dtype=np.int64)
        for k in range(min(tb.shape[0], max_objs)):
            length = tb[k, 3] / voxel_size[0] / out_size_factor
            width = tb[k, 4] / voxel_size[1] / out_size_factor
            if not (length > 0 and width > 0):
                continue
            radius = max(min_radius, int(gaussian_radius((length, width), min_overlap=gaussian_overlap)))
            ct = np.array([(tb[k, 0] - pc_range[0]) / voxel_size[0] / out_size_factor,
                           (tb[k, 1] - pc_range[1]) / voxel_size[1] / out_size_factor], dtype=np.float32)
            ct_int = ct.astype(np.int32)
            if not (0 <= ct_int[0] < fmap[0] and 0 <= ct_int[1] < fmap[1]):
                continue
            cls_id = int(tc[k]) - 1
            draw_gaussian(hm[cls_id], ct, radius)
            cat[k], ind[k], mask[k] = cls_id, ct_int[1] * fmap[0] + ct_int[0], 1
            anno[k] = np.concatenate((ct - ct_int, tb[k, 2:3], np.log(tb[k, 3:6]), tb[k, 6:8],
                                      np.sin(tb[k, -1:]), np.cos(tb[k, -1:])), axis=None)
        for key, val in (("hm", hm), ("anno_box", anno), ("ind", ind), ("mask", mask), ("cat", cat)):
            out[key].append(val)
        all_boxes.append(tb)
        all_classes.append(tc + first)
        first += n_cls
    flat_boxes = np.concatenate(all_boxes, 0) if all_boxes else np.zeros((0, 9), np.float32)
    flat_cls = np.concatenate(all_classes, 0) if all_classes else np.zeros(0, np.int32)
    if flat_boxes.shape[0] > max_objs:
        raise AssertionError("more objects (%d) than max_objs (%d)" % (flat_boxes.shape[0], max_objs))
    gbc = np.zeros((max_objs, 10), dtype=np.float32)
    gbc[: flat_boxes.shape[0]] = np.concatenate((flat_boxes[:, [0, 1, 2, 3, 4, 5, 8, 6, 7]],
                                                 flat_cls.reshape(-1, 1).astype(np.float32)), axis=1)
    out["gt_boxes_and_cls"] = gbc
    return out
