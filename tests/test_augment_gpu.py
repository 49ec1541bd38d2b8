"""GPU parity of the device augmentation pipeline (efg_amd/data/gpu_pipeline.py over csrc/augment.hip):
* against the golden vectors of the REFERENCE processors (same numpy seed -> same draws -> same kept points in the
  same order; coordinates to fp32 rounding of the reference's torch.matmul rotation, extra features bit-exact);
* against the oracle restatement, BIT-EXACT (identical fp32 operation order), on larger clouds."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import augment

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "augment_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_pipeline_matches_reference_golden(path):
    from efg_amd.data import gpu_pipeline as gp

    g = np.load(path)
    info = {"annotations": {"gt_boxes": g["boxes"].copy(), "labels": g["labels"].copy()}}
    np.random.seed(int(g["seed"]))
    pts, info = gp.run(gp.build_train_pipeline(g["pc_range"]), torch.from_numpy(g["points"]).cuda(), info)
    pts = pts.cpu().numpy()
    assert pts.shape == g["out_points"].shape
    np.testing.assert_allclose(pts, g["out_points"], rtol=2e-6, atol=2e-5)
    np.testing.assert_array_equal(pts[:, 3:], g["out_points"][:, 3:])
    np.testing.assert_array_equal(info["annotations"]["labels"], g["out_labels"])
    np.testing.assert_allclose(info["annotations"]["gt_boxes"], g["out_boxes"], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("n,f,seed", [(180000, 5, 1), (720000, 6, 2), (1, 5, 3), (1023, 4, 4), (1025, 3, 5)])
def test_bit_exact_vs_oracle(n, f, seed):
    from efg_amd.data import gpu_pipeline as gp
    from efg_amd.data.synthetic import PC_RANGE

    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, f)).astype(np.float32) * np.array([60, 60, 3] + [1] * (f - 3), np.float32)
    boxes = rng.normal(size=(7, 7)).astype(np.float32)
    np.random.seed(seed)
    prm = augment.draw_params(0.5, 0.78539816, 0.8, 1.2)
    ref_pts, _, _ = augment.pipeline(pts, boxes, prm, PC_RANGE)
    np.random.seed(seed)
    out, _ = gp.run(gp.build_train_pipeline(PC_RANGE), torch.from_numpy(pts).cuda(),
                    {"annotations": {"gt_boxes": boxes.copy(), "labels": np.arange(7)}})
    np.testing.assert_array_equal(out.cpu().numpy(), ref_pts)


def test_no_filter_and_empty():
    from efg_amd.data import gpu_pipeline as gp

    pts = np.random.default_rng(0).normal(size=(3000, 5)).astype(np.float32)
    dp = gp.DevicePoints(torch.from_numpy(pts).cuda())
    dp.queue(gp.TRANSLATE, 1.0, -2.0, 0.5)
    dp.queue(gp.SCALE, 2.0)
    out = dp.finalize().cpu().numpy()
    ref = pts.copy()
    ref[:, :3] += np.array([1.0, -2.0, 0.5], np.float32)
    ref[:, :3] *= np.float32(2.0)
    np.testing.assert_array_equal(out, ref)
    empty = gp.DevicePoints(torch.zeros((0, 5), device="cuda"))
    assert empty.materialize([-1, -1, -1, 1, 1, 1]) == 0
