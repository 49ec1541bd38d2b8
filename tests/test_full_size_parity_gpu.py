"""Parity at BASELINE's full scene size for the widened configurations: one training step of CenterPoint (1 x 180k
points) and of TrajectoryFormer (1 x 180k points, 60 objects, 11 frames of detector boxes) on the HIP path against
the SAME step with the oracle standing in for every HIP op on the host (same seed-0 weights, inputs and NumPy
generator).  The ConQueR twin of this check runs inside `bench.py` (`parity_full_size`).  ~40 s of host work."""
import os

import numpy as np
import pytest
import torch
from conftest import ROOT, force_proposals

pytestmark = pytest.mark.gpu


def _losses(make_trainer, make_batch, device, install, seed, forced_topk=None):
    np.random.seed(seed)
    tr = make_trainer(device)
    seen = {}
    transformer = getattr(tr.model, "transformer", None)
    if transformer is not None and hasattr(transformer, "_get_enc_proposals"):
        force_proposals(transformer, forced_topk)
        transformer.register_forward_hook(lambda mod, inp, out: seen.update(
            topk=mod.enc_outputs["topk_indexes"].detach().cpu()[..., 0], logits=mod.enc_outputs["pred_logits"].detach().cpu()[..., 0]))
    with install():
        loss_dict, total = tr.step(make_batch(device))
    out = {k: float(v.detach()) for k, v in loss_dict.items()}
    norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum().cpu() for p in tr.model.parameters() if p.grad is not None)))
    tr.close()
    return out, norm, seen


def _compare(make_trainer, make_batch, dev, oracle_mod, rel):
    import contextlib

    from oracle import cpu_backend

    torch.set_num_threads(16)
    cpu, cpu_norm, cpu_seen = _losses(make_trainer, make_batch, torch.device("cpu"), cpu_backend.install, 3)
    if "topk" in cpu_seen:
        # (1) the HIP path's OWN proposal choice may differ from the oracle path's only inside a tie: every token that is
        # in one set and not in the other scores within 2e-5 of the k-th best score (see Transformer._get_enc_proposals)
        _, _, own = _losses(make_trainer, make_batch, dev, contextlib.nullcontext, 3)
        for b in range(cpu_seen["topk"].shape[0]):
            sc, sg = set(cpu_seen["topk"][b].tolist()), set(own["topk"][b].tolist())
            kth = float(cpu_seen["logits"][b][cpu_seen["topk"][b]].min())
            for t in (sc ^ sg):
                assert abs(float(cpu_seen["logits"][b, t]) - kth) < 2e-5 and abs(float(own["logits"][b, t]) - kth) < 2e-5, \
                    "scene %d: token %d is in one proposal set only and is not part of the tie at the cut" % (b, t)
    # (2) the same proposals on both sides: every loss term and the gradient norm
    gpu, gpu_norm, _ = _losses(make_trainer, make_batch, dev, contextlib.nullcontext, 3, forced_topk=cpu_seen.get("topk"))
    assert set(cpu) == set(gpu)
    for k in cpu:
        assert gpu[k] == pytest.approx(cpu[k], rel=rel, abs=1e-6), k
    assert gpu_norm == pytest.approx(cpu_norm, rel=5e-4)


@pytest.mark.parametrize("sweeps,points", [(1, 180000), (4, 720000)])
def test_centerpoint_full_size_step(dev, oracle_mod, sweeps, points):
    """BASELINE configs[0] at the full scene size and configs[3]: the 4-sweep 720k x 6 cloud with the 200 000-voxel cap of
    `...36e.4f.improved/config.yaml` (the voxelizer's `break` path inside a training step)."""
    from efg_amd.centerpoint import VoxelNet
    from efg_amd.engine import Trainer, synthetic_batch

    cfg = os.path.join(ROOT, "configs", "centerpoint_waymo_voxelnet.yaml")
    ov = {}
    if sweeps > 1:
        ov = {"dataset.nsweeps": sweeps, "model.reader.num_input_features": 6, "model.backbone.num_input_features": 6,
              "dataset.processors.train.Voxelization.max_voxel_num": 200000,
              "dataset.processors.val.Voxelization.max_voxel_num": 400000}
    _compare(lambda d: Trainer(config=cfg, device=d, seed=0, model_cls=VoxelNet, ddp=False, max_iters=100, overrides=dict(ov)),
             lambda d: synthetic_batch(3000, 1, n_points=points, n_sweeps=sweeps, device=d if d.type == "cuda" else None),
             dev, oracle_mod, rel=2e-4)


def test_trajectoryformer_full_size_step(dev, oracle_mod):
    from efg_amd.engine import Trainer
    from efg_amd.tracking import TrajectoryFormer
    from efg_amd.tracking.synthetic import synthetic_tracking_batch

    cfg = os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml")
    _compare(lambda d: Trainer(config=cfg, device=d, seed=0, model_cls=TrajectoryFormer, ddp=False, max_iters=100),
             lambda d: synthetic_tracking_batch(7000, 1, device=d if d.type == "cuda" else None, n_points=180000,
                                                n_objects=60, n_false=20), dev, oracle_mod, rel=2e-4)


@pytest.mark.parametrize("model,queries,scenes", [("conquer", 1000, 1), ("conquer", 900, 1), ("voxeldetr", 1000, 2)])
def test_conquer_full_size_step(dev, oracle_mod, model, queries, scenes):
    """The headline configurations themselves (BASELINE.json configs[1], configs[2]): 180k-point scenes, ConQueR with the
    YAML's 1000 queries and with BASELINE's 900, Voxel-DETR at its batch of 2; every loss term + the gradient norm."""
    from efg_amd.engine import Trainer, synthetic_batch

    cfg = None if model == "conquer" else os.path.join(ROOT, "configs", "voxeldetr_waymo_res18.yaml")

    def trainer(d):
        tr = Trainer(config=cfg, device=d, seed=0, ddp=False, overrides={"model.transformer.num_queries": queries})
        tr.model.noise_generator = torch.Generator().manual_seed(4321)      # the CDN noise, drawn on the host
        return tr

    _compare(trainer, lambda d: synthetic_batch(1000, scenes, n_points=180000, device=d if d.type == "cuda" else None),
             dev, oracle_mod, rel=1e-4)
