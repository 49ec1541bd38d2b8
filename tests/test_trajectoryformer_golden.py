"""TrajectoryFormer (BASELINE configs[4]) training step vs the reference's own `forward_train`
(tests/golden/trajectoryformer_small.npz, made by scripts/make_golden_trajectoryformer.py from the reference code
imported in place).  Same deterministic weights, same inputs, same NumPy generator seed on both sides.

CPU variant: our host code over the oracle's rotated IoU / NMS (oracle/cpu_backend.py).  GPU variant: the product
path -- HIP IoU / NMS kernels, everything else on PyTorch-ROCm.

Tolerances: integer / boolean results (masks, kept boxes, point selections) exact; fp32 activations 5e-5 of the
tensor's scale; losses 1e-4 relative; gradients 1e-3 of their maximum (sums over 1e4..1e5 fp32 terms in a different
order than the reference's [token, batch]-major attention).
"""
import os

import numpy as np
import pytest
import torch
from conftest import ROOT, golden
from golden_init import deterministic_state, tracking_inputs

FIXTURE = "trajectoryformer_small.npz"


def _build(device):
    from efg_amd.config import load_config
    from efg_amd.tracking import TrajectoryFormer

    cfg = load_config(os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"),
                      {"model.device": str(device)})
    torch.manual_seed(0)
    model = TrajectoryFormer(cfg)
    model.load_state_dict(deterministic_state(model.state_dict()))
    model.train()
    return model


def _close(name, got, want, tol):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) if got.size else 0.0
    assert err <= tol * scale, "%s: max err %.3e (scale %.3g, tol %.1e)" % (name, err, scale, tol)


def _exact(name, got, want):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, "%s differs in %d places, first at %s: %s vs %s" % (
        name, len(bad), bad[0].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])


def _run(model, g, grad_tol):
    import efg_amd.tracking.geometry as geo
    import efg_amd.tracking.trajectoryformer as tfm

    seen = {}
    for name in ("organize_proposals", "hypotheses_augment", "generate_trajectory_hypothses",
                 "get_trajcetory_point_feature", "get_trajectory_boxes_feature", "get_trajectory_hypotheses_feat",
                 "get_cls_targets", "get_reg_targets"):
        fn = getattr(model, name)

        def wrapped(*a, _fn=fn, _name=name, **k):
            r = _fn(*a, **k)
            seen[_name] = r
            return r

        setattr(model, name, wrapped)
    crop = tfm.crop_current_frame_points

    def crop_wrapped(*a, **k):
        seen["crop"] = crop(*a, **k)
        return seen["crop"]

    tfm.crop_current_frame_points = crop_wrapped
    try:
        batch = tracking_inputs()
        for i, (sample, info) in enumerate(batch):                    # the generator is deterministic; make sure
            for k in ("gt_boxes", "pred_boxes3d", "pred_scores", "pred_labels"):
                _exact("in.%s.%d" % (k, i), info["annotations"][k], g["in.%s.%d" % (k, i)])
            assert float(np.abs(sample[0]["points"]).sum(dtype=np.float64)) == float(g["in.points_abs_sum"][i])
        np.random.seed(1234)
        losses = model(batch)
    finally:
        tfm.crop_current_frame_points = crop
    _exact("numpy generator state after the step", np.random.get_state()[1][:8].astype(np.int64), g["rng_after"])

    # proposals -> trajectories -> hypotheses: pure selection + a little fp32 geometry
    for i, tol in enumerate((5e-5, 0, 0, 0)):
        got, want = seen["organize_proposals"][i], g["organize_proposals.%d" % i]
        _close("organize_proposals.%d" % i, got, want, tol) if tol else _exact("organize_proposals.%d" % i, got, want)
    _close("hypotheses_augment", seen["hypotheses_augment"], g["hypotheses_augment"], 1e-6)
    for i in range(2):
        _close("hypotheses.%d" % i, seen["generate_trajectory_hypothses"][i], g["generate_trajectory_hypothses.%d" % i],
               5e-5)
    # same points in the same slots; an EMPTY ROI is filled with its own (fp32, forecast-derived) centre, hence not `==`
    _close("cropped points", seen["crop"], g["crop_current_frame_points"], 1e-6)
    for i in range(3):
        _close("point token %d" % i, seen["get_trajcetory_point_feature"][i],
               g["get_trajcetory_point_feature.%d" % i], 5e-5)
    _close("box-sequence feature", seen["get_trajectory_boxes_feature"], g["get_trajectory_boxes_feature"], 5e-5)
    _close("hypothesis feature", seen["get_trajectory_hypotheses_feat"], g["get_trajectory_hypotheses_feat"], 5e-5)
    _exact("fg_iou_mask", seen["get_cls_targets"][0], g["get_cls_targets.0"])
    _exact("fg_reg_mask", seen["get_cls_targets"][1], g["get_cls_targets.1"])
    _close("ious_targets", seen["get_cls_targets"][2], g["get_cls_targets.2"], 5e-5)
    _exact("gt_boxes", seen["get_cls_targets"][3], g["get_cls_targets.3"])
    _close("reg_targets", seen["get_reg_targets"], g["get_reg_targets"], 5e-5)
    assert bool(g["get_cls_targets.0"].any()) and bool(g["get_cls_targets.1"].any())      # the fixture has foreground
    for k in ("loss_cls", "loss_reg"):
        assert float(losses[k].detach()) == pytest.approx(float(g["loss." + k]), rel=1e-4), k

    sum(v.sum() for v in losses.values()).backward()
    params = dict(model.named_parameters())
    assert sorted(n for n, p in params.items() if p.grad is None) == sorted(g["no_grad"].tolist())
    for key in g:
        if key.startswith("grad."):
            rows = g["rows." + key[5:]]
            got = params[key[5:]].grad.detach().cpu().double().numpy()
            full_scale = float(g["gradmax." + key[5:]])
            err = float(np.abs(got[rows] - g[key]).max())
            assert err <= grad_tol * full_scale, "%s: %.3e vs scale %.3e" % (key, err, full_scale)
            assert float(np.abs(got).max()) == pytest.approx(full_scale, rel=1e-2)
    _close("BatchNorm running mean", model.seqboxembed.feat.bn1.running_mean, g["bn_running_mean"], 5e-5)


def test_trajectoryformer_forward_train_matches_reference_cpu(oracle_mod):
    from oracle import cpu_backend

    torch.set_num_threads(8)
    with cpu_backend.install():
        _run(_build("cpu"), golden(FIXTURE), 1e-3)


@pytest.mark.gpu
def test_trajectoryformer_forward_train_matches_reference_gpu(dev):
    _run(_build(dev), golden(FIXTURE), 1e-3)


def test_trajectoryformer_trainer_step_cpu(oracle_mod):
    """The experiment's solver block (AdamW, OneCycle, gradient-norm clipping) around the model, through the same
    Trainer as the detectors; degenerate input (no proposals) gives the reference's zero losses."""
    from oracle import cpu_backend

    from efg_amd.engine import Trainer
    from efg_amd.tracking import TrajectoryFormer
    from efg_amd.tracking.synthetic import synthetic_tracking_batch

    torch.set_num_threads(8)
    np.random.seed(5)
    tr = Trainer(config=os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"), device="cpu",
                 model_cls=TrajectoryFormer, ddp=False, max_iters=10)
    assert tr.grad_clipper is not None and tr.lr_scheduler is not None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.step(synthetic_tracking_batch(40, 1, n_points=3000, n_objects=4, n_false=1))
    before = tr.model.point_reg.layers[0].weight.detach().clone()
    with cpu_backend.install():
        loss_dict, total = tr.step(synthetic_tracking_batch(40, 2, n_points=6000, n_objects=6, n_false=2))
    assert set(loss_dict) == {"loss_cls", "loss_reg"} and torch.isfinite(total)
    assert not torch.equal(before, tr.model.point_reg.layers[0].weight)
    assert not tr.model.velboxembed.training and tr.model.seqboxembed.training       # the forecast module stays frozen
    tr.close()


def _ragged_detections(seed, widths, t1=11):
    rng = np.random.default_rng(seed)
    boxes, scores, labels = [], [], []
    for m in widths:
        centre = rng.uniform(-20, 20, (m, 2))
        tab = np.zeros((t1, m, 9), np.float32)
        for f in range(t1):
            tab[f, :, 0:2] = centre + rng.normal(0, 0.15, (m, 2)) + (rng.uniform(size=(m, 1)) < 0.3) * rng.normal(0, 0.6, (m, 2))
            tab[f, :, 2] = rng.normal(-0.8, 0.1, m)
            tab[f, :, 3:6] = rng.uniform(0.6, 4.5, (m, 3))
            tab[f, :, 6:8] = rng.normal(0, 1.0, (m, 2))
            tab[f, :, 8] = rng.uniform(-np.pi, np.pi, m)
            tab[f, 1::5] = tab[f, 0::5][: len(tab[f, 1::5])]          # near-duplicates: NMS has something to suppress
            tab[f, 1::5, 0] += 0.05
        sc = rng.uniform(0.05, 0.95, (t1, m)).astype(np.float32)
        drop = rng.uniform(size=(t1, m)) < 0.25                      # zero padding inside frames, as the loader leaves it
        tab[drop] = 0
        sc[drop] = 0
        boxes.append(tab.reshape(-1, 9))
        scores.append(sc.reshape(-1))
        labels.append(rng.integers(1, 4, t1 * m).astype(np.float32) * (~drop.reshape(-1)))
    return boxes, scores, labels


def _organisation_case(model, device):
    """Ragged scenes (different widths, duplicated / overlapping boxes, zero padding): the one-launch organisation must
    equal the reference's per-frame loop, and the scene-batched linking a per-scene loop."""
    import efg_amd.tracking.trajectoryformer as tfm

    widths = (9, 4, 14)
    boxes, scores, labels = _ragged_detections(3, widths)
    model.batch_size = len(widths)
    to = lambda xs: [torch.from_numpy(x).to(device) for x in xs]  # noqa: E731
    frames_a, labels_a = model._organize_batched(to(boxes), to(scores), to(labels), max(widths))
    frames_b, labels_b = model._organize_loop(to(boxes), to(scores), to(labels))
    assert frames_a.shape == frames_b.shape and frames_a.shape[2] < max(widths) + 1
    # kept boxes are the same sets in the same order except among exact score ties of all-zero padding rows
    assert torch.equal(frames_a, frames_b) and torch.equal(labels_a, labels_b)
    assert int((frames_a[..., 3:6].sum(-1) > 0).sum()) < sum(11 * w for w in widths) * 0.75      # NMS removed boxes
    traj, valid = model.generate_trajectory(frames_a[:, 1:])
    for s in range(len(widths)):                                     # per-scene restatement of the linking
        prop = frames_a[s, 1:]
        cur = prop[0]
        for i in range(1, prop.shape[0]):
            moved = torch.cat([cur[:, 0:2] - 0.1 * cur[:, 6:8], cur[:, 2:]], -1)
            iou = tfm.boxes_iou3d_gpu(moved[:, [0, 1, 2, 3, 4, 5, -1]], prop[i][:, [0, 1, 2, 3, 4, 5, -1]])
            best, arg = iou.max(1)
            cur = torch.where((best >= 0.5)[:, None], prop[i][arg], torch.zeros_like(cur))
            assert torch.equal(traj[s, i], cur) and torch.equal(valid[s, i], best >= 0.5)
    assert bool(valid[:, 1:].any()) and not bool(valid[:, 1:].all())


def test_batched_organisation_equals_reference_loop_cpu(oracle_mod):
    from oracle import cpu_backend

    with cpu_backend.install():
        _organisation_case(_build("cpu"), "cpu")


@pytest.mark.gpu
def test_batched_organisation_equals_reference_loop_gpu(dev):
    _organisation_case(_build(dev), dev)


def _edge_batches():
    from efg_amd.tracking.synthetic import make_tracking_sample

    a = make_tracking_sample(900, n_points=20000, n_objects=6, n_false=2)
    b = make_tracking_sample(901, n_points=20000, n_objects=5, n_false=1)
    no_gt = make_tracking_sample(902, n_points=20000, n_objects=4, n_false=2)
    for k in ("gt_boxes", "labels", "difficulty", "num_points_in_gt"):
        no_gt[1]["annotations"][k] = no_gt[1]["annotations"][k][:0]
    no_points = make_tracking_sample(903, n_points=20000, n_objects=4, n_false=2)
    no_points[0][0]["points"] = no_points[0][0]["points"][:0]
    return {"one scene without ground truth": [a, no_gt], "one scene without points": [b, no_points]}


def _edge_case(model, install):
    """Edge inputs of the reference's branches (:768-802 no ground truth in a scene; utils.py:415-420 no points in a
    ROI): finite losses, gradients for every trained parameter."""
    for name, batch in _edge_batches().items():
        np.random.seed(7)
        model.zero_grad(set_to_none=True)
        with install():
            losses = model(batch)
            total = sum(v.sum() for v in losses.values())
            total.backward()
        assert torch.isfinite(total), name
        assert model.point_reg.layers[0].weight.grad is not None and torch.isfinite(model.point_reg.layers[0].weight.grad).all()


def test_trajectoryformer_edge_inputs_cpu(oracle_mod):
    from oracle import cpu_backend

    _edge_case(_build("cpu"), cpu_backend.install)


@pytest.mark.gpu
def test_trajectoryformer_edge_inputs_gpu(dev):
    import contextlib

    _edge_case(_build(dev), contextlib.nullcontext)
