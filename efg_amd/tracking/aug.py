"""TrajectoryFormer's loader-side augmentations on device clouds: counterparts of `$TF/aug.py`
(`CusTomRandomFlip3D`, `CusTomGlobalRotation`, `CusTomGlobalScaling`, `CusTomFilterByRange`; `$TF/config.yaml:30-41`).

They are the detection pipeline's processors (efg_amd/data/gpu_pipeline.py: the cloud stays in HBM, the per-point
arithmetic is queued and executed by one kernel) extended the way the reference extends them: the detector boxes the
tracker consumes (`annotations["pred_boxes3d"]`, [n, 9] = x y z l w h vx vy heading) are flipped / rotated / scaled
with the ground truth, and the range filter leaves every `pred*` / `future*` entry alone.  Same class names, arguments
and NumPy generator calls as the reference."""
import numpy as np

from ..data import gpu_pipeline as base
from ..data.gpu_pipeline import _all_annotations, _rotate_z


class CusTomRandomFlip3D(base.RandomFlip3D):
    """aug.py:58-118."""

    def __call__(self, points, info):
        before = np.random.get_state()
        points, info = super().__call__(points, info)
        if "annotations" not in info or "pred_boxes3d" not in info["annotations"]:
            return points, info
        after = np.random.get_state()
        np.random.set_state(before)                       # the same two draws decide what happened to the cloud
        flip_y = np.random.choice([False, True], replace=False, p=[1 - self.p, self.p])
        flip_x = np.random.choice([False, True], replace=False, p=[1 - self.p, self.p])
        np.random.set_state(after)
        pred = info["annotations"]["pred_boxes3d"]
        if flip_y:
            pred[:, 1] = -pred[:, 1]
            pred[:, -1] = -pred[:, -1]
            pred[:, 7] = -pred[:, 7]
        if flip_x:
            pred[:, 0] = -pred[:, 0]
            pred[:, -1] = -(pred[:, -1] + np.pi)
            pred[:, 6] = -pred[:, 6]
        return points, info


class CusTomGlobalRotation(base.GlobalRotation):
    """aug.py:121-178."""

    def __call__(self, points, info):
        state = np.random.get_state()
        noise_rotation = np.random.uniform(self.rotation[0], self.rotation[1])
        np.random.set_state(state)
        points, info = super().__call__(points, info)      # draws the same angle
        for ann in _all_annotations(info):
            if "pred_boxes3d" in ann:
                pred = ann["pred_boxes3d"]
                pred[:, :3] = _rotate_z(pred[:, :3], noise_rotation)
                pred[:, -1] += noise_rotation
                if pred.shape[1] > 7:
                    vel = np.hstack([pred[:, 6:8], np.zeros((pred.shape[0], 1), pred.dtype)])
                    pred[:, 6:8] = _rotate_z(vel, noise_rotation)[:, :2]
        return points, info


class CusTomGlobalScaling(base.GlobalScaling):
    """aug.py:181-200 (the sweeps' annotations carry no detector boxes there either)."""

    def __call__(self, points, info):
        state = np.random.get_state()
        noise_scale = np.random.uniform(self.min_scale, self.max_scale)
        np.random.set_state(state)
        points, info = super().__call__(points, info)
        if "annotations" in info and "pred_boxes3d" in info["annotations"]:
            info["annotations"]["pred_boxes3d"][:, :-1] *= noise_scale
        return points, info


def _dict_select(dict_, inds):
    """aug.py:17-27: ground-truth entries follow the kept boxes, detector / future entries are left alone."""
    for k, v in dict_.items():
        if "pred" in k or "future" in k:
            continue
        if isinstance(v, dict):
            _dict_select(v, inds)
        else:
            dict_[k] = v[inds]


class CusTomFilterByRange(base.FilterByRange):
    """aug.py:30-55."""

    def __init__(self, pc_range, with_gt=True):
        super().__init__(pc_range, with_gt=with_gt, with_data=True)

    def __call__(self, points, info):
        points.materialize(self.pc_range)
        if self.with_gt:
            for ann in _all_annotations(info):
                _dict_select(ann, self._box_mask(ann["gt_boxes"]))
        return points, info
