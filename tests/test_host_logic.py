"""CPU: host-side logic that mirrors the reference's Python (no device ops involved)."""
import numpy as np
import pytest
import torch


def _ref_dn_mask(pad_size, single_pad, dn_number, num_queries):
    """Literal restatement of the mask loops at $CQ/cdn.py:98-112 (True = blocked)."""
    tgt = pad_size + num_queries
    m = torch.zeros(tgt, tgt, dtype=torch.bool)
    m[pad_size:, :pad_size] = True
    m[:pad_size, pad_size:] = True
    for i in range(dn_number):
        lo, hi = single_pad * 2 * i, single_pad * 2 * (i + 1)
        if i == 0:
            m[lo:hi, hi:pad_size] = True
        if i == dn_number - 1:
            m[lo:hi, :lo] = True
        else:
            m[lo:hi, hi:pad_size] = True
            m[lo:hi, :lo] = True
    return m


@pytest.mark.parametrize("single_pad,dn_number,nq", [(5, 3, 11), (1, 1, 4), (7, 2, 3), (0, 3, 5)])
def test_dn_attention_mask_matches_reference_loops(single_pad, dn_number, nq):
    from efg_amd.detection3d.cdn import dn_attn_mask

    pad = single_pad * 2 * dn_number
    got = dn_attn_mask(pad, single_pad, dn_number, nq, torch.device("cpu"))
    assert torch.equal(got, _ref_dn_mask(pad, single_pad, dn_number, nq))


def test_cdn_layout_and_noise_bounds():
    from efg_amd.detection3d.cdn import prepare_for_cdn

    g = torch.Generator().manual_seed(0)
    targets = [{"labels": torch.tensor([0, 2, 1]), "gt_boxes": torch.rand(3, 7, generator=g) * 0.5 + 0.25},
               {"labels": torch.tensor([1]), "gt_boxes": torch.rand(1, 7, generator=g) * 0.5 + 0.25}]
    lab, box, mask, meta = prepare_for_cdn((targets, 3, 0.5, 0.4), True, 20, 3, 256, None, generator=g)
    assert meta == {"pad_size": 18, "num_dn_group": 3}
    assert lab.shape == (2, 18, 3) and box.shape == (2, 18, 7) and mask.shape == (38, 38)
    # scene 1 has one GT: slots 1,2 of every half-group stay zero padding
    assert (box[1, 1:3] == 0).all() and (box[1, 0] != 0).any()
    assert ((box >= 0) & (box <= 1)).all()
    # positive copies stay within +-half-size*scale of the GT centre, negatives move further on average
    pos = (box[0, 0:3, :3] - targets[0]["gt_boxes"][:, :3]).abs()
    assert (pos <= targets[0]["gt_boxes"][:, 3:6] * 0.5 * 0.4 + 1e-6).all()


def test_box_coder_roundtrip_and_range():
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.detection3d.box_coder import VoxelBoxCoder3D

    _, boxes, labels = make_scene(3, n_points=1000)
    coder = VoxelBoxCoder3D(VOXEL_SIZE, PC_RANGE)
    tgt = coder.encode({"gt_boxes": torch.from_numpy(boxes.copy()), "labels": torch.from_numpy(labels.copy())})
    assert tgt["gt_boxes"].shape[1] == 7 and tgt["labels"].min() >= 0 and tgt["labels"].max() <= 2
    dec = coder.decode(tgt["gt_boxes"].clone())
    np.testing.assert_allclose(dec[:, :6].numpy(), boxes[:, :6], rtol=1e-4, atol=1e-3)
    dyaw = (dec[:, 6].numpy() - boxes[:, 8] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(dyaw).max() < 1e-3


def test_giou_and_matcher_small_case():
    from efg_amd.detection3d.matcher import HungarianMatcher3d
    from efg_amd.detection3d.utils import box_cxcyczlwh_to_xyxyxy, generalized_box3d_iou, paired_box3d_giou

    b = torch.tensor([[0.5, 0.5, 0.5, 0.2, 0.2, 0.2], [0.2, 0.2, 0.5, 0.1, 0.1, 0.1]])
    g = generalized_box3d_iou(box_cxcyczlwh_to_xyxyxy(b), box_cxcyczlwh_to_xyxyxy(b))
    torch.testing.assert_close(torch.diag(g), torch.ones(2))
    assert g[0, 1] < 0  # disjoint boxes
    torch.testing.assert_close(paired_box3d_giou(box_cxcyczlwh_to_xyxyxy(b), box_cxcyczlwh_to_xyxyxy(b.flip(0))),
                               torch.stack([g[0, 1], g[1, 0]]))
    m = HungarianMatcher3d(1, 4, 2, 4)
    logits = torch.full((1, 4, 3), -4.0)
    logits[0, 2, 1] = 4.0
    logits[0, 0, 0] = 4.0
    boxes = torch.rand(1, 4, 7) * 0.2 + 0.4
    tg = [{"labels": torch.tensor([1, 0]), "gt_boxes": torch.stack([boxes[0, 2], boxes[0, 0]])}]
    (qi, gi), = m({"pred_logits": logits, "pred_boxes": boxes}, tg)
    assert dict(zip(gi.tolist(), qi.tolist())) == {0: 2, 1: 0}


def test_config_loader_interpolation(tmp_path):
    from efg_amd.config import load_config

    p = tmp_path / "c.yaml"
    p.write_text("a:\n  b: [1, 2]\n  c: 3\nd: ${a.b}\ne:\n  f: ${a}\n")
    cfg = load_config(str(p), {"a.c": 9})
    assert cfg.d == [1, 2] and cfg.e.f.c == 9 and cfg.a.c == 9
    with pytest.raises(AttributeError):
        cfg.missing


def test_synthetic_scene_is_deterministic_and_in_range():
    from efg_amd.data.synthetic import PC_RANGE, make_scene

    a = make_scene(2003, n_points=30000)
    b = make_scene(2003, n_points=30000)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    p = a[0]
    assert p.shape == (30000, 5) and p.dtype == np.float32
    assert (p[:, 0] >= PC_RANGE[0]).all() and (p[:, 0] < PC_RANGE[3]).all() and (p[:, 2] >= PC_RANGE[2]).all()
    assert make_scene(5, n_points=9000, n_sweeps=4)[0].shape[1] == 6


def test_batched_contrastive_loss_matches_reference_loop():
    """Our batched evaluation vs a literal restatement of the Python loops at $CQ/voxel_detr.py:223-254."""
    from efg_amd.engine import Trainer

    tr = Trainer(device="cpu", overrides={"model.transformer.num_queries": 30, "model.transformer.enc_layers": 1},
                 seed=0, ddp=False)
    m = tr.model
    g = torch.Generator().manual_seed(0)
    nq, groups, n_layers = 30, 3, 3
    per_gt = [4, 2]
    max_gt = max(per_gt)
    tot = nq + (groups + 1) * max_gt
    oc = torch.randn(n_layers, 2, tot, 3, generator=g)
    ob = torch.rand(n_layers, 2, tot, 7, generator=g)
    targets = [{"gt_boxes": torch.rand(n, 7)} for n in per_gt]
    matched = [(torch.tensor([5, 17, 2, 9]), torch.tensor([0, 1, 2, 3])), (torch.tensor([11, 0]), torch.tensor([1, 0]))]
    q_of_g = torch.full((2, max_gt), -1, dtype=torch.int64)
    for bi, (qi, gi) in enumerate(matched):
        q_of_g[bi, gi] = qi
    got = m._contrastive_losses(oc, ob, q_of_g, targets, {"num_dn_group": groups})
    sim_f = torch.nn.CosineSimilarity(dim=2)
    num_gts = sum(per_gt)
    for li in range(n_layers):
        loss = 0.0
        projs = torch.cat((oc[li], ob[li]), dim=-1)
        gt_projs = m.projector(projs[:, nq:].detach())
        pred_projs = m.predictor(m.projector(projs[:, :nq]))
        pos_idxs = list(range(1, groups + 1))
        for bi, idx in enumerate(matched):
            sim = sim_f(gt_projs[bi].unsqueeze(1), pred_projs[bi].unsqueeze(0)) / m.tau
            pairs = torch.stack(idx, dim=-1)
            neg_mask = projs.new_ones(nq).bool()
            neg_mask[pairs[:, 0]] = False
            for pair in pairs:
                pos_mask = torch.tensor([int(pair[1] + max_gt * pi) for pi in pos_idxs])
                pos_pair = sim[pos_mask, pair[0]].view(-1, 1)
                neg_pairs = sim[:, neg_mask][pos_mask]
                loss = loss + (torch.log(torch.exp(pos_pair) + torch.exp(neg_pairs).sum(dim=-1, keepdim=True))
                               - pos_pair).mean()
        ref = m.contras_loss_coeff * loss / num_gts
        assert float(got[f"loss_contrastive_dec_{li}"]) == pytest.approx(float(ref), rel=1e-5)


def test_bench_bare_multi_gpu_command_names_the_missing_devices():
    """`python bench.py --gpus 8` (no launcher) on a node with fewer devices: a clear message, not a rendezvous hang or
    an assert about torchrun (the bare command starts its own ranks when the devices exist; GPU test in
    test_multirank_gpu.py)."""
    import os
    import subprocess
    import sys

    import torch

    from conftest import ROOT

    n = torch.cuda.device_count() + 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EFG_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "needs %d devices" % n in r.stderr, r.stderr[-2000:]
