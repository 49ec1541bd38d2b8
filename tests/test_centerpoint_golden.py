"""CenterPoint (BASELINE configs[0] / [3]) against the REFERENCE's own `VoxelNet.forward` ($CP1/voxelnet.py:194-226):
tests/golden/centerpoint_full_small.npz is produced by scripts/make_golden_centerpoint.py from the reference model
imported in place (label assignment :43-187, SpMiddleResNetFHD wiring sparse_net.py:473-545, RPN
configurable_rpn.py:14-122, CenterHead + FastFocalLoss / RegLoss center_head.py:104-171, backward).  Weights / inputs
are regenerated from tests/golden_init.py.  CPU: oracle ops (this is also BASELINE configs[0]: the CenterPoint graph,
batch of small clouds, no GPU).  GPU: the HIP path from raw points + the inference branch with the HIP rotated NMS."""
import contextlib
import copy
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from golden_init import CENTERPOINT_OVERRIDES, deterministic_state, full_inputs


def _build(device):
    from efg_amd.centerpoint import VoxelNet
    from efg_amd.config import load_config

    ov = dict(CENTERPOINT_OVERRIDES)
    ov["model.device"] = str(device)
    cfg = load_config(os.path.join(ROOT, "configs", "centerpoint_waymo_voxelnet.yaml"), ov)
    torch.manual_seed(0)
    model = VoxelNet(cfg)
    model.load_state_dict({k: v.to(device) for k, v in deterministic_state(model.state_dict()).items()}, strict=True)
    model.train()
    return model, cfg, golden("centerpoint_full_small.npz")


def _batch(device, cfg, names=True):
    points_list, annos = full_inputs()
    cls = np.array(cfg.dataset.classes)
    out = []
    for p, a in zip(points_list, annos):
        a = copy.deepcopy(a)
        if names:
            a["gt_names"] = cls[a["labels"] - 1]
        out.append(({"points": torch.from_numpy(p).to(device)}, {"annotations": a}))
    return out


def _check(model, g, cap, losses, grad_tol):
    c = lambda x: x.detach().float().cpu().numpy()  # noqa: E731
    for t in range(1):  # label assignment: bit-for-bit the reference's arrays
        for key in ("hm", "anno_box", "ind", "mask", "cat"):
            np.testing.assert_array_equal(c(cap["targets"][key][t]).astype(g["tgt::%s::%d" % (key, t)].dtype),
                                          g["tgt::%s::%d" % (key, t)], err_msg=key)
    for name, got, want in (("BEV", c(cap["bev"]), g["bev"]), ("RPN", c(cap["rpn"])[:, ::8], g["rpn_sub"])):
        err = np.abs(got - want).max() / max(np.abs(want).max(), 1.0)
        assert err <= 5e-5, "%s map: %.2e" % (name, err)
    ref = {k[6:]: float(v) for k, v in g.items() if k.startswith("loss::")}
    assert set(ref) == set(losses)
    for k, v in ref.items():
        assert float(losses[k]) == pytest.approx(v, rel=2e-4, abs=1e-5), k
    params = dict(model.named_parameters())
    assert int(g["n_params"]) == sum(p.numel() for p in params.values())
    # Gradients: the fixture also records how far the REFERENCE's own fp32 gradient is from the same model run in
    # fp64 (`graderr64::*`, scripts/make_golden_centerpoint.py).  Weight gradients behind a BatchNorm are
    # ill-conditioned (1-2e-2 for the backbone / RPN here; a bias in front of a BatchNorm has an exactly-zero
    # gradient and carries only noise), so the bar is max(grad_tol, 2 x the reference's own error).
    for k, v in g.items():
        if k.startswith("grad::"):
            own = float(g["graderr64::" + k[6:]])
            if own > 1.0:
                continue
            got = c(params[k[6:]].grad)
            got = got[:8] if got.size > 65536 else got
            err = np.abs(got - v).max() / np.abs(v).max()
            assert err <= max(grad_tol, 2 * own), "%s: %.2e (reference fp32 vs fp64: %.2e)" % (k, err, own)


def _run(model, cfg, device):
    cap = {}
    hooks = [model.backbone.register_forward_hook(lambda m, i, o: cap.update(bev=o.detach())),
             model.neck.register_forward_hook(lambda m, i, o: cap.update(rpn=o.detach()))]
    orig = model.label_assign

    def spy(*a, **k):
        cap["targets"] = orig(*a, **k)
        return cap["targets"]

    model.label_assign = spy
    losses = model(_batch(device, cfg))
    total = sum(v for k, v in losses.items() if k.endswith("_loss") and v.requires_grad)
    total.backward()
    for h in hooks:
        h.remove()
    return cap, losses


def test_reference_centerpoint_cpu(oracle_mod):
    from oracle import cpu_backend

    torch.set_num_threads(8)
    model, cfg, g = _build(torch.device("cpu"))
    with cpu_backend.install():
        cap, losses = _run(model, cfg, torch.device("cpu"))
    _check(model, g, cap, losses, grad_tol=1e-3)


def test_label_assignment_accepts_integer_labels():
    """Synthetic scenes carry `labels` 1..K instead of `gt_names`: same targets."""
    from efg_amd.centerpoint.targets import assign_scene

    _, annos = full_inputs()
    tasks = [{"num_classes": 3, "class_names": ["VEHICLE", "PEDESTRIAN", "CYCLIST"]}]
    names = np.array(tasks[0]["class_names"])
    kw = dict(tasks=tasks, class_names_plain=list(names), grid_size=np.array([128, 128, 40]),
              pc_range=[-6.4, -6.4, -2.0, 6.4, 6.4, 4.0], voxel_size=[0.1, 0.1, 0.15], out_size_factor=8,
              gaussian_overlap=0.1, max_objs=20, min_radius=2)
    a = assign_scene(dict(annos[0], gt_names=names[annos[0]["labels"] - 1]), **kw)
    b = assign_scene({"gt_boxes": annos[0]["gt_boxes"], "labels": annos[0]["labels"]}, **kw)
    for key in ("hm", "anno_box", "ind", "mask", "cat"):
        np.testing.assert_array_equal(a[key][0], b[key][0])
    assert int(a["mask"][0].sum()) == len(annos[0]["labels"]) and float(a["hm"][0].max()) == 1.0


@pytest.mark.gpu
def test_reference_centerpoint_gpu(dev):
    model, cfg, g = _build(dev)
    cap, losses = _run(model, cfg, dev)
    torch.cuda.synchronize()
    _check(model, g, cap, losses, grad_tol=1e-3)
    # inference branch: decode + HIP rotated NMS, reference output format
    model.eval()
    with torch.no_grad():
        res = model(_batch(dev, cfg))
    assert len(res) == 2 and set(res[0]) == {"scores", "labels", "boxes3d"}
    assert res[0]["boxes3d"].shape[1] == 7 and res[0]["scores"].shape[0] <= 300
    assert res[0]["scores"].numel() == 0 or (float(res[0]["scores"].min()) > 0.1 and int(res[0]["labels"].min()) >= 1)


def _check_inference(model, g, cfg, device, install):
    """Inference branch vs the reference's own eval-mode forward (decode, masks, rotated NMS in pcdet convention,
    truncation): same kept boxes in the same order."""
    model.eval()
    with torch.no_grad(), install():
        res = model(_batch(device, cfg))
    assert len(res) == 2
    for i, r in enumerate(res):
        want_scores = g["infer::scores::%d" % i]
        assert 10 < len(want_scores) <= 60                     # suppression happened, truncation bounds it
        assert r["scores"].shape[0] == len(want_scores), (r["scores"].shape, len(want_scores))
        np.testing.assert_array_equal(r["labels"].numpy(), g["infer::labels::%d" % i])
        np.testing.assert_allclose(r["scores"].numpy(), want_scores, rtol=0, atol=1e-5)
        np.testing.assert_allclose(r["boxes3d"].numpy(), g["infer::boxes3d::%d" % i], rtol=0, atol=1e-4)


def test_reference_centerpoint_inference_cpu(oracle_mod):
    from oracle import cpu_backend

    torch.set_num_threads(8)
    model, cfg, g = _build(torch.device("cpu"))
    _check_inference(model, g, cfg, torch.device("cpu"), cpu_backend.install)


@pytest.mark.gpu
def test_reference_centerpoint_inference_gpu(dev):
    model, cfg, g = _build(dev)
    _check_inference(model, g, cfg, dev, contextlib.nullcontext)


def _head_pair(device, dtype):
    """Two copies of one SepHead ($CP1/center_head.py:18-52) with non-trivial BatchNorm parameters, an input map and
    output gradients."""
    from efg_amd.centerpoint import center_head as ch

    torch.manual_seed(0)
    heads = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "hm": (3, 2)}
    a = ch.SepHead(64, heads, bn="BN", final_kernel=3).to(device=device, dtype=dtype)
    for m in a.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_()
    b = copy.deepcopy(a)
    x = torch.randn(2, 64, 24, 20, device=device, dtype=dtype)
    if device.type == "cuda":
        a, b = a.to(memory_format=torch.channels_last), b.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    gos = {k: torch.randn(2, v[0], 24, 20, device=device, dtype=dtype) for k, v in heads.items()}
    return ch, a, b, x, gos


def _head_fused_against_stacks(device, dtype, tol):
    ch, fused, plain, x, gos = _head_pair(device, dtype)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    assert fused._fusable(xa)
    saved = ch._HEAD_FUSE
    try:
        ch._HEAD_FUSE = True
        ya = fused(xa)
        ch._HEAD_FUSE = False
        yb = plain(xb)
    finally:
        ch._HEAD_FUSE = saved
    assert list(ya) == list(yb)
    torch.autograd.backward([ya[k] for k in ya], [gos[k] for k in ya])
    torch.autograd.backward([yb[k] for k in yb], [gos[k] for k in yb])

    def close(u, v, what):
        err = float((u.double() - v.double()).abs().max()) / max(float(v.double().abs().max()), 1e-30)
        assert err <= tol, "%s: %.2e" % (what, err)

    for k in ya:
        assert ya[k].shape == yb[k].shape
        close(ya[k], yb[k], k)
    close(xa.grad, xb.grad, "input gradient")
    pa, pb = dict(fused.named_parameters()), dict(plain.named_parameters())
    for k in pa:   # every parameter of every stack gets ITS block of the joint gradient
        assert pa[k].grad is not None and pa[k].grad.shape == pa[k].shape, k
        if k.endswith(".0.bias"):   # a bias in front of a BatchNorm: its gradient is exactly zero, both forms carry rounding noise
            scale = float(pb[k[:-4] + "weight"].grad.abs().max())
            assert float(pa[k].grad.abs().max()) <= 1e-4 * scale and float(pb[k].grad.abs().max()) <= 1e-4 * scale, k
            continue
        close(pa[k].grad, pb[k].grad, k)
    ba, bb = dict(fused.named_buffers()), dict(plain.named_buffers())
    for k in ba:   # running statistics and num_batches_tracked land in the modules' own buffers
        close(ba[k], bb[k], k)
    assert int(ba["hm.1.num_batches_tracked"]) == 1
    # eval mode: running statistics, same maps
    fused.eval(), plain.eval()
    with torch.no_grad():
        try:
            ch._HEAD_FUSE = True
            ea = fused(x)
            ch._HEAD_FUSE = False
            eb = plain(x)
        finally:
            ch._HEAD_FUSE = saved
    for k in ea:
        close(ea[k], eb[k], "eval " + k)


def test_task_stacks_as_one_stack_cpu():
    """SepHead's five conv + BN + ReLU + conv stacks evaluated as ONE 320-wide stack (block-diagonal second weight) are the
    five stacks: outputs, input gradient, every parameter's gradient, the BatchNorm buffers; float64 so that only the
    summation order differs."""
    _head_fused_against_stacks(torch.device("cpu"), torch.float64, 1e-11)


@pytest.mark.gpu
def test_task_stacks_as_one_stack_gpu(dev):
    """The same on the GPU path (MIOpen convolutions on channels-last maps, the HIP BatchNorm over 320 channels)."""
    _head_fused_against_stacks(dev, torch.float32, 2e-5)
