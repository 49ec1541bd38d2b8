"""CPU: the oracle's restatement of the reference augmentation chain (oracle/augment.py) against golden vectors
produced by the REFERENCE classes themselves (scripts/make_golden_augment.py: RandomFlip3D -> GlobalRotation ->
GlobalScaling -> FilterByRange -> PointShuffle of efg/data/augmentations/extend_3d.py, imported in place).
Same numpy seed => same random draws => same kept points in the same order; coordinates agree to fp32 rounding of
the rotation (the reference uses torch.matmul)."""
import glob
import os

import numpy as np
import pytest

from oracle import augment

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "augment_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_chain_matches_reference(path):
    g = np.load(path)
    np.random.seed(int(g["seed"]))
    prm = augment.draw_params(0.5, 0.78539816, 0.8, 1.2)
    pts, boxes, keep = augment.pipeline(g["points"], g["boxes"], prm, g["pc_range"])
    assert pts.shape == g["out_points"].shape          # identical filter decisions
    np.testing.assert_allclose(pts, g["out_points"], rtol=2e-6, atol=2e-5)   # and identical order (shuffle)
    np.testing.assert_array_equal(pts[:, 3:], g["out_points"][:, 3:])        # untouched features: bit-exact
    np.testing.assert_array_equal(g["labels"][keep], g["out_labels"])
    np.testing.assert_allclose(boxes, g["out_boxes"], rtol=2e-6, atol=2e-5)


def test_flip_and_scale_are_bit_exact_without_rotation():
    g = np.load(GOLDEN[0])
    prm = {"flip_x_axis": True, "flip_y_axis": True, "angle": 0.0, "scale": 1.1}
    pts = augment.transform_points(g["points"], prm)
    ref = g["points"].copy()
    ref[:, 1] = -ref[:, 1]
    ref[:, 0] = -ref[:, 0]
    ref[:, :3] *= np.float32(1.1)
    np.testing.assert_array_equal(pts, ref)


def test_range_mask_is_inclusive():
    r = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    p = np.array([[1, 1, 1, 0], [-1, -1, -1, 0], [1.0000001, 0, 0, 0], [0, np.nan, 0, 0]], np.float32)
    assert augment.mask_points_by_range(p, r).tolist() == [True, True, False, False]
