"""TrajectoryFormer's loader-side augmentations vs the reference's `$TF/aug.py` (tests/golden/tf_augment.npz,
scripts/make_golden_tf_aug.py): flip / rotate / scale / range-filter of the cloud, the ground truth AND the detector
boxes (`pred_boxes3d`), for three seeds."""
import numpy as np
import pytest
import torch
from conftest import golden

PC_RANGE = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]


def _case(case):
    from efg_amd.tracking.synthetic import make_tracking_sample

    sample, info = make_tracking_sample(600 + case, n_points=20000, n_objects=12, n_false=4)
    pts = sample[0]["points"].copy()
    pts[:, :3] *= 1.2
    info["annotations"]["gt_boxes"][:, :2] *= 1.35
    info["sweeps"] = []
    return pts, info


def _chain():
    from efg_amd.tracking import aug

    return [aug.CusTomRandomFlip3D(p=0.5), aug.CusTomGlobalRotation(rotation=0.78539816),
            aug.CusTomGlobalScaling(min_scale=0.95, max_scale=1.05), aug.CusTomFilterByRange(pc_range=PC_RANGE)]


class _NoCloud:
    """Stands in for DevicePoints on a box without a GPU: the annotation arithmetic and the generator protocol are
    host-side and can be checked alone."""

    def queue(self, *a):
        pass

    def materialize(self, *a):
        return 0


def _check_annotations(g, case, info, tol):
    ann = info["annotations"]
    for k in ("gt_boxes", "pred_boxes3d"):
        np.testing.assert_allclose(ann[k], g["case%d.%s" % (case, k)], rtol=0, atol=tol)
    np.testing.assert_array_equal(ann["labels"], g["case%d.labels" % case])
    np.testing.assert_array_equal(ann["pred_scores"], g["case%d.pred_scores" % case])


def test_annotations_and_generator_match_reference_cpu():
    g = golden("tf_augment.npz")
    for case in range(3):
        _, info = _case(case)
        np.random.seed(int(g["case%d.seed" % case]))
        pts = _NoCloud()
        for proc in _chain():
            pts, info = proc(pts, info)
        _check_annotations(g, case, info, 2e-5)
    np.testing.assert_array_equal(np.random.get_state()[1][:8].astype(np.int64), g["rng_after"])


@pytest.mark.gpu
def test_cloud_and_annotations_match_reference_gpu(dev):
    from efg_amd.data.gpu_pipeline import DevicePoints

    g = golden("tf_augment.npz")
    for case in range(3):
        pts, info = _case(case)
        np.random.seed(int(g["case%d.seed" % case]))
        cloud = DevicePoints(torch.from_numpy(pts).to(dev))
        for proc in _chain():
            cloud, info = proc(cloud, info)
        out = cloud.finalize().cpu().numpy()
        want = g["case%d.points" % case]
        assert out.shape == want.shape                                  # the same points survive the range filter
        np.testing.assert_allclose(out, want, rtol=0, atol=2e-5)
        _check_annotations(g, case, info, 2e-5)
