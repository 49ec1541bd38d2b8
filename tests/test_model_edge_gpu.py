"""GPU: edge-case batches through the whole training step (voxelizer -> backbone -> transformer -> device matcher ->
fused losses -> backward -> AdamW)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trainer():
    from efg_amd.engine import Trainer

    return Trainer(device=torch.device("cuda:0"), seed=0)


def _strip(sample):
    ann = sample[1]["annotations"]
    for k in list(ann):
        ann[k] = ann[k][:0]


@pytest.mark.parametrize("case", ["normal", "one_scene_without_gt", "single_gt", "tiny_cloud_batch1"])
def test_step_runs_and_is_finite(trainer, case):
    from efg_amd.engine import synthetic_batch

    dev = torch.device("cuda:0")
    if case == "single_gt":
        batch = synthetic_batch(5300, 2, device=dev, n_boxes=1)
    elif case == "tiny_cloud_batch1":
        batch = synthetic_batch(5400, 1, device=dev, n_points=2000)
    else:
        batch = synthetic_batch(5000, 2, device=dev)
        if case == "one_scene_without_gt":
            _strip(batch[0])
    loss_dict, total = trainer.step(batch)
    assert len(loss_dict) == 32 and bool(torch.isfinite(total))
    assert all(torch.isfinite(p.grad).all() for p in trainer.model.parameters() if p.grad is not None)


def test_batch_without_any_gt_fails_like_the_reference(trainer):
    """$CQ/transformer.py:149 divides by max_gt_num * 2: a batch with no GT box at all raises ZeroDivisionError in the
    reference; the mirror keeps that behaviour (the Waymo loader never produces such a batch)."""
    from efg_amd.engine import synthetic_batch

    batch = synthetic_batch(5200, 2, device=torch.device("cuda:0"))
    for s in batch:
        _strip(s)
    with pytest.raises(ZeroDivisionError):
        trainer.step(batch)
