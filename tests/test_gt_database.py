"""DatabaseSampling with the object database resident in HBM vs the reference processor
(tests/golden/gt_database.npz, scripts/make_golden_gt_database.py): four consecutive samples from one seed -- which
objects are pasted (class quotas, shuffled cursors that wrap, BEV collision rejection incl. crossing and containment),
the annotation updates, the generator state, and (GPU) the pasted cloud bit for bit."""
import numpy as np
import pytest
import torch
from conftest import golden


def _database(device):
    from efg_amd.data.gt_database import DatabaseSampling, DeviceGTDatabase
    from efg_amd.data.synthetic_db import make_database

    infos, clouds = make_database(seed=7)
    np.random.seed(99)
    db = DeviceGTDatabase(infos, clouds, [{"VEHICLE": 15}, {"PEDESTRIAN": 10}, {"CYCLIST": 10}], min_points=5,
                          difficulty=-1, device=device)
    return DatabaseSampling(db)


def test_collision_test_semantics():
    from efg_amd.data.gt_database import bev_corners, box_collision_test

    boxes = np.array([[0, 0, 4, 2, 0.3], [0.2, 0.1, 1, 0.5, 1.0], [1.5, 0.5, 4, 2, -0.4], [10, 10, 2, 2, 0.0],
                      [12.0, 10, 2, 2, 0.0], [12.05, 10, 2, 2, 0.0]], np.float32)
    c = bev_corners(boxes[:, :2], boxes[:, 2:4], boxes[:, 4])
    m = box_collision_test(c, c)
    assert m[0, 1] and m[1, 0]            # containment without an edge crossing
    assert m[0, 2]                        # crossing edges
    assert not m[0, 3] and not m[3, 4]    # apart / touching edges only (hull overlap must be strict)
    assert not m[4, 5]                    # (reference behaviour) equal axis-aligned boxes shifted along an axis: collinear
    #                                       edges do not "cross" and no corner is STRICTLY inside


def test_selection_and_annotations_match_reference_cpu():
    from efg_amd.data.synthetic_db import make_sampling_scene

    g = golden("gt_database.npz")
    proc = _database(None)
    pasted_rows = []
    for case in range(4):
        pts, info = make_sampling_scene(300 + case)
        out, info = proc(torch.from_numpy(pts), info)     # a host tensor: the selection logic is device-independent
        ann = info["annotations"]
        np.testing.assert_array_equal(ann["gt_boxes"].astype(np.float32), g["case%d.gt_boxes" % case])
        assert [str(n) for n in ann["gt_names"]] == g["case%d.gt_names" % case].tolist()
        np.testing.assert_array_equal(np.asarray(ann["num_points_in_gt"], np.int64), g["case%d.num_points_in_gt" % case])
        np.testing.assert_array_equal(out.numpy(), g["case%d.points" % case])
        pasted_rows.append(out.shape[0] - pts.shape[0])
    np.testing.assert_array_equal(np.random.get_state()[1][:8].astype(np.int64), g["rng_after"])
    assert min(pasted_rows) > 1000


@pytest.mark.gpu
def test_pasted_cloud_matches_reference_gpu(dev):
    from efg_amd.data.gpu_pipeline import DevicePoints
    from efg_amd.data.synthetic_db import make_sampling_scene

    g = golden("gt_database.npz")
    proc = _database(dev)
    assert proc.db_sampler.points.is_cuda
    for case in range(4):
        pts, info = make_sampling_scene(300 + case)
        out, info = proc(DevicePoints(torch.from_numpy(pts).to(dev)), info)
        cloud = out.finalize()
        assert cloud.is_cuda
        np.testing.assert_array_equal(cloud.cpu().numpy(), g["case%d.points" % case])
        np.testing.assert_array_equal(info["annotations"]["gt_boxes"].astype(np.float32), g["case%d.gt_boxes" % case])


@pytest.mark.gpu
def test_device_pipeline_feeds_the_training_step(dev):
    """The whole loader-side chain of $CQ/config.yaml:24-42 on the device -- ground-truth paste from the HBM-resident
    database, flip, rotation, scaling, range filter, shuffle -- straight into one ConQueR training step: no host copy of
    the cloud anywhere between the raw sweep and the loss."""
    from efg_amd.data.gpu_pipeline import DevicePoints, build_train_pipeline, run
    from efg_amd.data.synthetic import PC_RANGE, make_scene
    from efg_amd.data.synthetic_db import make_database
    from efg_amd.engine import Trainer

    from efg_amd.data.gt_database import DeviceGTDatabase

    np.random.seed(5)
    infos, clouds = make_database(seed=7)
    db = DeviceGTDatabase(infos, clouds, [{"VEHICLE": 15}, {"PEDESTRIAN": 10}, {"CYCLIST": 10}], min_points=5, device=dev)
    chain = build_train_pipeline(PC_RANGE, database=db)
    names = np.array(["VEHICLE", "PEDESTRIAN", "CYCLIST"])
    batch = []
    for s in range(2):
        pts, boxes, labels = make_scene(9000 + s, n_points=40000, n_boxes=8)
        info = {"annotations": {"gt_boxes": boxes[:, [0, 1, 2, 3, 4, 5, 8]].copy(), "gt_names": names[labels - 1],
                                "difficulty": np.zeros(len(labels), np.int64),
                                "num_points_in_gt": np.full(len(labels), 50, np.int64)}}
        cloud, info = run(chain, DevicePoints(torch.from_numpy(pts).to(dev)), info)
        ann = info["annotations"]
        assert cloud.is_cuda and len(ann["gt_boxes"]) > 8 and cloud.shape[0] > 30000      # objects were pasted
        ann["labels"] = np.array([list(names).index(n) + 1 for n in ann["gt_names"]], np.int64)
        ann["gt_boxes"] = ann["gt_boxes"].astype(np.float32)
        batch.append(({"points": cloud}, {"annotations": ann}))
    tr = Trainer(device=dev, seed=0, overrides={"model.transformer.num_queries": 100, "model.transformer.enc_layers": 1},
                 ddp=False)
    loss_dict, total = tr.step(batch)
    assert torch.isfinite(total) and len(loss_dict) == 32
    tr.close()
