"""CPU (no GPU): BASELINE config 0 -- a 16k-point synthetic cloud, batch 1 -- steps through the whole
model with the oracle standing in for every HIP op (oracle/cpu_backend.py).  This exercises the host
logic (collate layout, sparse ResNet wiring, FPN, transformer, CDN, matcher, 32 losses, backward,
AdamW param groups) without a GPU, and checks that the product refuses to run on CPU on its own."""
import numpy as np
import pytest
import torch


def _trainer(**ov):
    from efg_amd.engine import Trainer

    overrides = {"model.transformer.num_queries": 60, "model.transformer.enc_layers": 1}
    overrides.update(ov)
    return Trainer(device="cpu", overrides=overrides, seed=0, ddp=False)


def _batch(n_scenes=1, n_points=16000, seed=1000):
    from efg_amd.engine import synthetic_batch

    return synthetic_batch(seed, n_scenes, n_points=n_points, n_boxes=6)


def test_product_has_no_cpu_fallback():
    tr = _trainer()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.step(_batch(n_points=2000))


def test_config0_train_step_on_cpu(oracle_mod):
    from oracle import cpu_backend

    torch.set_num_threads(8)
    tr = _trainer()
    names = {n for n, _ in tr.model.named_parameters()}
    assert "backbone.extractor.bottom_up.stem.conv1.0.weight" in names          # reference state-dict names
    assert "transformer.encoder.layers.0.self_attn.linear_box_weight" in names
    assert tr.model.backbone.extractor.bottom_up.stem.conv1[0].weight.shape == (16, 3, 3, 3, 5)  # [Cout,kd,kh,kw,Cin]
    with cpu_backend.install():
        loss_dict, total = tr.step(_batch())
    assert torch.isfinite(total)
    keys = set(loss_dict)
    assert {"loss_ce", "loss_bbox", "loss_giou", "loss_rad", "loss_ce_enc", "loss_ce_dn", "loss_ce_0", "loss_ce_dn_1",
            "loss_contrastive_dec_2", "accuracy"} <= keys
    assert len(keys) == 32  # same 32 terms as the reference model
    # parameters of branches that feed nothing get no gradient -- exactly the reference's dead set
    dead = sorted({n.split(".")[2] if n.startswith("backbone.extractor.") else n
                   for n, p in tr.model.named_parameters() if p.requires_grad and p.grad is None})
    assert dead == ["bottom_up", "fpn_lateral2", "fpn_output2", "fpn_output4"], dead
    no_grad_bottom = {n for n, p in tr.model.named_parameters() if p.requires_grad and p.grad is None
                      and "bottom_up" in n}
    assert all("res2_out" in n for n in no_grad_bottom)


def test_full_reference_graph_matches_pruned_graph(oracle_mod):
    """Evaluating the unused FPN levels (reference behaviour) changes no loss and no gradient.  (Reduced grid of the
    full-model golden: the statement is about the graph, not the scene size.)"""
    import copy

    from golden_init import FULL_OVERRIDES, full_inputs
    from oracle import cpu_backend

    torch.set_num_threads(8)
    points_list, annos = full_inputs()
    res = []
    for full in (False, True):
        ov = dict(FULL_OVERRIDES)
        if full:
            ov["model.eval_unused_levels"] = True
        tr = _trainer(**ov)
        tr.model.noise_generator = torch.Generator().manual_seed(5)
        batch = [({"points": torch.from_numpy(p)}, {"annotations": copy.deepcopy(a)}) for p, a in zip(points_list, annos)]
        with cpu_backend.install():
            tr.optimizer.zero_grad()
            ld = tr.model(batch)
            total = sum(v for v in ld.values() if v.requires_grad)
            total.backward()
        g = tr.model.backbone.extractor.bottom_up.stem.conv1[0].weight.grad.clone()
        res.append((float(total), g))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-6)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-5, atol=1e-7)


def test_reference_input_format_is_accepted(oracle_mod):
    """Samples voxelized on the host (the reference's DataLoader format, extend_3d.py:267-283) go
    through `collate` and give the same losses as raw points voxelized by the op."""
    from oracle import cpu_backend

    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE

    torch.set_num_threads(8)
    batch = _batch(n_points=5000)
    pts = batch[0][0]["points"].numpy()
    v, c, n = oracle_mod.hard_voxelize(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
    ref_sample = {"voxels": v, "coordinates": c, "num_points_per_voxel": n, "points": pts,
                  "num_voxels": np.array([v.shape[0]], np.int64), "shape": np.array([1504, 1504, 40], np.int64),
                  "range": np.array(PC_RANGE, np.float32), "size": np.array(VOXEL_SIZE, np.float32)}
    out = []
    for b in (batch, [(ref_sample, batch[0][1])]):
        tr = _trainer()
        tr.model.noise_generator = torch.Generator().manual_seed(3)
        with cpu_backend.install():
            ld = tr.model(b)
        out.append(float(sum(v for v in ld.values() if v.requires_grad)))
    assert out[0] == pytest.approx(out[1], rel=1e-5)


def test_config0_centerpoint_train_step_on_cpu(oracle_mod):
    """BASELINE configs[0] as named: the CenterPoint graph (mean reader -> SpMiddleResNetFHD -> RPN -> CenterHead),
    one 16k-point synthetic cloud, batch 1, no GPU -- forward, the reference's loss dict, backward, AdamW + OneCycle +
    gradient clipping, through the same Trainer the GPU path uses."""
    from oracle import cpu_backend

    from efg_amd.centerpoint import VoxelNet
    from efg_amd.engine import Trainer

    torch.set_num_threads(8)
    import os

    from conftest import ROOT

    tr = Trainer(config=os.path.join(ROOT, "configs", "centerpoint_waymo_voxelnet.yaml"), device="cpu",
                 model_cls=VoxelNet, ddp=False, max_iters=10)
    assert tr.grad_clipper is not None and tr.lr_scheduler is not None      # the experiment's solver block
    names = {n for n, _ in tr.model.named_parameters()}
    assert {"backbone.conv_input.0.weight", "neck.blocks.1.4.weight", "neck.deblocks.1.0.weight",
            "center_head.shared_conv.0.weight", "center_head.tasks.0.hm.3.bias"} <= names   # reference state-dict names
    with cpu_backend.install():
        loss_dict, total = tr.step(_batch(n_points=16000))
    assert set(loss_dict) == {"0_loss", "0_hm_loss", "0_loc_loss", "0_num_positive"} and torch.isfinite(total)
    assert float(loss_dict["0_num_positive"]) == 6.0
    assert all(p.grad is not None for p in tr.model.parameters())            # find_unused_parameters: False holds
    tr.close()


def test_res50_backbone_has_the_reference_module_tree():
    """depth 50 = bottleneck blocks (reference sparse_net.py:168-237, :380-388): [3, 4, 6, 3] blocks per stage, each
    `conv.{0,3,6}` = 1x1x1 / 3x3x3 / 1x1x1 convolutions with norms at `conv.{1,4,7}`, `shortcut.{0,1}` where the width
    changes, bottleneck width doubling per stage -- the parameter names and shapes a reference state dict carries."""
    from efg_amd.modeling.backbones.sparse_net import SparseBottleneckBlock, build_sparse_resnet_backbone

    cfg = dict(depth=50, norm="BN1d", activation=dict(type="ReLU", inplace=True), stem_out_channels=32,
               out_features=["res3", "res4"], num_groups=1, width_per_group=16, res1_out_channels=64)
    net = build_sparse_resnet_backbone(cfg, 5)
    sd = net.state_dict()
    assert [len(getattr(net, "res%d" % s)) for s in (2, 3, 4)] == [3, 4, 6]
    assert all(isinstance(b, SparseBottleneckBlock) for s in (2, 3, 4) for b in getattr(net, "res%d" % s))
    # spconv weight layout [cout, kd, kh, kw, cin]; bottleneck widths 16 / 32 / 64, stage widths 64 / 128 / 256
    assert tuple(sd["res2.0.conv.0.weight"].shape) == (16, 1, 1, 1, 32)
    assert tuple(sd["res2.0.conv.3.weight"].shape) == (16, 3, 3, 3, 16)
    assert tuple(sd["res2.0.conv.6.weight"].shape) == (64, 1, 1, 1, 16)
    assert tuple(sd["res2.0.conv.7.weight"].shape) == (64,)
    assert tuple(sd["res2.0.shortcut.0.weight"].shape) == (64, 3, 3, 3, 32)
    assert "res2.1.shortcut.0.weight" not in sd
    assert tuple(sd["res4.0.conv.3.weight"].shape) == (64, 3, 3, 3, 64)
    assert net.res3[0].stride == 2 and net.res3[1].stride == 1
