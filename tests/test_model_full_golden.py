"""Full-model parity against the REFERENCE's own `VoxelDETR.forward` ($CQ/voxel_detr.py:118-256), SURVEY.md §4
item 5 / §8c row 5: tests/golden/conquer_full_small.npz is produced by scripts/make_golden_full.py from the
reference model imported in place (its SparseResNet wiring sparse_net.py:284-309, FPN.forward fpn.py:136-169,
input_proj, Transformer, CDN, matcher, 32 losses incl. the contrastive double loop, backward), with the numba
voxelizer of the reference on the input side.  Weights and inputs are regenerated on both sides from
tests/golden_init.py; only activations / losses / gradients are stored.

CPU variant: the oracle stands in for the HIP ops (pins the host logic + the oracle's sparse conv to the dense
stand-in the reference ran on).  GPU variant: the real HIP path from raw points (voxelizer, rulebooks, MFMA sparse
conv, BEV flatten, fused box attention, device matcher, fused losses), pruned and full graph.
"""
import contextlib
import copy
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from golden_init import FULL_OVERRIDES, deterministic_state, full_inputs


def _build(device, full_graph, fixture="conquer_full_small.npz", yaml_name="conquer_waymo_res18.yaml", extra=None):
    from efg_amd.config import load_config
    from efg_amd.detection3d.voxel_detr import VoxelDETR

    ov = dict(FULL_OVERRIDES)
    ov["model.device"] = str(device)
    if full_graph:
        ov["model.eval_unused_levels"] = True
    ov.update(extra or {})
    cfg = load_config(os.path.join(ROOT, "configs", yaml_name), ov)
    torch.manual_seed(0)
    model = VoxelDETR(cfg)
    state = deterministic_state(model.state_dict())
    model.load_state_dict({k: v.to(device) for k, v in state.items()}, strict=True)  # reference names and shapes
    model.train()
    return model, golden(fixture)


def _run(model, device):
    import efg_amd.detection3d.voxel_detr as vd

    points_list, annos = full_inputs()
    batch = [({"points": torch.from_numpy(p).to(device)}, {"annotations": copy.deepcopy(a)})
             for p, a in zip(points_list, annos)]
    cap = {}
    ext = model.backbone.extractor
    hooks = [
        ext.bottom_up.register_forward_hook(lambda m, i, o: cap.update({"bu_" + k: v.detach() for k, v in o.items()})),
        ext.register_forward_hook(lambda m, i, o: cap.update({"fpn_" + k: v.detach() for k, v in o.items()})),
        model.transformer.register_forward_pre_hook(lambda m, a: cap.update(src=a[0][0].detach())),
        model.transformer.register_forward_hook(lambda m, i, o: cap.update(memory=o[3].detach(), topk=o[5].detach())),
    ]
    orig = vd.prepare_for_cdn

    def spy(*a, **k):
        r = orig(*a, **k)
        cap.update(dn_label=r[0].detach().cpu(), dn_box=r[1].detach().cpu())
        return r

    vd.prepare_for_cdn = spy
    try:
        model.noise_generator = torch.Generator().manual_seed(1234)  # the reference run's CDN stream (CPU generator)
        losses = model(batch)
        total = sum(v for v in losses.values() if v.requires_grad)
        total.backward()
    finally:
        vd.prepare_for_cdn = orig
        for h in hooks:
            h.remove()
    return cap, losses, total


# gradients that do not pass through a sampling location or a BatchNorm (the 1e-4 tier; scripts/grad_tier_report.py prints the
# measured error of every stored gradient against the reference's float64 run: profiles/r06_grad_tier_table.txt)
EXACT_GRADS = ("detection_head", "proposal_head", "decoder.layers.2.norm", "decoder.layers.2.linear2",
               "decoder.layers.2.multihead_attn.out_proj", "decoder.layers.2.multihead_attn.value_proj",
               "decoder.layers.2.self_attn", "projector", "predictor")


def _check(model, g, cap, losses, total, full_graph, act_tol=5e-5, loss_rtol=1e-4, grad_tol=3e-3, grad_tol_exact=1e-4):
    c = lambda x: x.detach().float().cpu().numpy()  # noqa: E731

    def close(name, got, want, tol):
        scale = max(float(np.abs(want).max()), 1.0)
        err = float(np.abs(got - want).max())
        assert err <= tol * scale, "%s: max abs err %.3e (scale %.3g, tol %.1e)" % (name, err, scale, tol)

    # the denoising queries: same generator stream => the reference's noised boxes / labels, value for value
    if "dn_box" in g:
        np.testing.assert_allclose(c(cap["dn_box"]), g["dn_box"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(c(cap["dn_label"]), g["dn_label"])
    # sparse backbone wiring + BEV flatten (channel = c * D + d), FPN top-down, input projection, encoder
    if "bu_res3" in g:
        close("res3 BEV", c(cap["bu_res3"]), g["bu_res3"], act_tol)
        close("res4 BEV", c(cap["bu_res4"]), g["bu_res4"], act_tol)
        if full_graph:
            close("res2 BEV (every 8th channel)", c(cap["bu_res2"])[:, ::8], g["bu_res2_sub"], act_tol)
            close("p2 (every 16th channel)", c(cap["fpn_p2"])[:, ::16], g["fpn_p2_sub"], act_tol)
        close("p3", c(cap["fpn_p3"]), g["fpn_p3"], act_tol)
        close("src", c(cap["src"]), g["src"], act_tol)
    close("memory", c(cap["memory"]), g["memory"], act_tol)
    assert all(set(a) == set(b) for a, b in zip(c(cap["topk"])[..., 0], g["topk"][..., 0])), "different proposals"
    ref_losses = {k[6:]: v for k, v in g.items() if k.startswith("loss::")}
    assert set(ref_losses) == set(losses), set(ref_losses) ^ set(losses)
    assert len(ref_losses) == (32 if "loss_contrastive_dec_0" in ref_losses else 17)
    for k, v in sorted(ref_losses.items()):
        np.testing.assert_allclose(float(losses[k]), float(v), rtol=loss_rtol, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(float(total), float(g["total_loss"]), rtol=1e-4)
    params = dict(model.named_parameters())
    assert int(g["n_params"]) == sum(p.numel() for p in params.values() if p.requires_grad)
    # Gradients.  Two tiers (see the module docstring of scripts/grad_bisect.py and DESIGN.md "bilinear kinks"):
    #  * tensors whose gradient does not pass through a sampling location (the heads and the last decoder layer's
    #    biases / norms, evaluated before any sampling backward): 1e-4 of the tensor's max;
    #  * everything behind a box-attention backward: 3e-3.  Bilinear sampling is piecewise linear in the location;
    #    the reference run has ~350 000 sampling coordinates and the closest lies 9.5e-7 px from an integer
    #    (fixture key `min_kink_distance_px`), inside the fp32 rounding of a pixel coordinate, so CPU and GPU pick
    #    different one-sided derivatives for a handful of samples.  Measured on MI355X: ONE query row differs (1 % of
    #    its own gradient), every other row agrees to 1e-7; that row's contribution is ~1e-3 of a weight gradient --
    #    with the HIP kernels and with PyTorch's own grid_sample on the GPU alike.  Convolutions in front of a
    #    BatchNorm are ill-conditioned on top: the reference's own fp32 gradient is 6-8e-4 off its fp64 run.
    exact = EXACT_GRADS
    for k, v in g.items():
        if k.startswith("grad::"):
            got = c(params[k[6:]].grad)
            if got.size > 65536:
                got = got[:8]
            tol = grad_tol_exact if any(e in k for e in exact) else grad_tol
            err = float(np.abs(got - v).max() / np.abs(v).max())
            assert err <= tol, "%s: gradient error %.2e of the tensor max (tolerance %.1e)" % (k, err, tol)
    dead_ref = set(str(g["dead_params"]).split(";"))
    dead = {n for n, p in params.items() if p.requires_grad and p.grad is None}
    assert dead == dead_ref, dead ^ dead_ref
    bn = model.backbone.extractor.bottom_up.stem.conv1[1]
    np.testing.assert_allclose(c(bn.running_mean), g["bn_running_mean_after"], rtol=1e-4, atol=1e-6)


def test_reference_full_model_cpu(oracle_mod):
    from oracle import cpu_backend

    torch.set_num_threads(8)
    model, g = _build(torch.device("cpu"), full_graph=True)
    with cpu_backend.install():
        cap, losses, total = _run(model, torch.device("cpu"))
    # same PyTorch CPU arithmetic as the reference run around the (oracle) ops: no kink flips, 1e-3 everywhere
    _check(model, g, cap, losses, total, full_graph=True, grad_tol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("full_graph", [False, True])
def test_reference_full_model_gpu(dev, full_graph):
    model, g = _build(dev, full_graph)
    cap, losses, total = _run(model, dev)
    torch.cuda.synchronize()
    _check(model, g, cap, losses, total, full_graph)


def _against_fp64(model):
    """Every stored gradient against the reference model's FLOAT64 run (tests/golden/conquer_full_small_grad64.npz,
    scripts/make_golden_full.py --grad64-out): (ours-vs-64, ref32-vs-64) per tensor, as max |diff| / max |g64|."""
    g64 = golden("conquer_full_small_grad64.npz")
    params = dict(model.named_parameters())
    rows = {}
    for k in g64:
        if not k.startswith("grad64::"):
            continue
        name = k[8:]
        got = params[name].grad.detach().double().cpu().numpy()
        if got.size > 65536:
            got = got[:8]
        want = g64[k]
        rows[name] = (float(np.abs(got - want).max() / np.abs(want).max()), float(g64["ref32_err::" + name]))
    return rows


def _check_fp64(rows):
    """The gradient bar stated against the TRUTH instead of against the reference's fp32 run: the 1e-4 tier for every tensor
    that does not sit behind a sampling location or a BatchNorm; for the others the old hard bound (one bilinear corner
    9.5e-7 px from an integer may flip with the last bit of the geometry: 1e-3 of a tensor) AND their median at 1e-4 -- a
    flip moves one tensor chain, not the median.  Measured: every tensor <= 6e-6 on the CPU (oracle ops), <= 1.7e-5 on the
    GPU, i.e. CLOSER to the float64 run than the reference's own fp32 gradients are (6.4e-4 in front of a BatchNorm);
    profiles/r06_grad_tier_table.txt."""
    assert len(rows) >= 30
    loose = []
    for name, (err, ref_err) in rows.items():
        if any(e in name for e in EXACT_GRADS):
            assert err <= 1e-4, "%s: %.2e of the tensor max off the float64 reference run" % (name, err)
        else:
            assert err <= 3e-3, "%s: %.2e of the tensor max off the float64 reference run" % (name, err)
            loose.append(err)
    assert float(np.median(loose)) <= 1e-4, "median error of the BatchNorm / sampling tier %.2e" % float(np.median(loose))


def test_gradients_against_the_reference_float64_run_cpu(oracle_mod):
    from oracle import cpu_backend

    torch.set_num_threads(8)
    model, g = _build(torch.device("cpu"), full_graph=True)
    with cpu_backend.install():
        _run(model, torch.device("cpu"))
    _check_fp64(_against_fp64(model))


@pytest.mark.gpu
def test_gradients_against_the_reference_float64_run_gpu(dev):
    model, g = _build(dev, True)
    _run(model, dev)
    torch.cuda.synchronize()
    _check_fp64(_against_fp64(model))


VD = dict(fixture="voxeldetr_full_small.npz", yaml_name="voxeldetr_waymo_res18.yaml")


def test_reference_voxeldetr_variant_cpu(oracle_mod):
    """BASELINE config 1 by name: plain Voxel-DETR ($VD/voxel_detr.py: no momentum decoder, no denoising queries, no
    contrastive loss) against the reference's own VoxelDETR of that experiment directory."""
    from oracle import cpu_backend

    torch.set_num_threads(8)
    model, g = _build(torch.device("cpu"), full_graph=False, **VD)
    names = {n for n, _ in model.named_parameters()}
    assert not any(n.startswith(("transformer.decoder_gt", "projector", "predictor")) for n in names)
    with cpu_backend.install():
        cap, losses, total = _run(model, torch.device("cpu"))
    _check(model, g, cap, losses, total, full_graph=False)
    model.eval()
    with cpu_backend.install(), torch.no_grad():
        points_list, _ = full_inputs()
        res = model([({"points": torch.from_numpy(points_list[0])}, {})])
    assert res[0]["scores"].shape[0] == min(300, 30 * 3) and res[0]["boxes3d"].shape[1] == 7  # top-300 rule


@pytest.mark.gpu
def test_reference_voxeldetr_variant_gpu(dev):
    model, g = _build(dev, full_graph=False, **VD)
    cap, losses, total = _run(model, dev)
    torch.cuda.synchronize()
    _check(model, g, cap, losses, total, full_graph=False)


# ---- inference branch ------------------------------------------------------------------------------------------------
INFER = {"conquer": ("conquer_infer_small.npz", "conquer_waymo_res18.yaml"),
         "voxeldetr": ("voxeldetr_infer_small.npz", "voxeldetr_waymo_res18.yaml")}


def _check_inference(which, device, install):
    """Eval-mode forward vs the reference model's own (scripts/make_golden_full.py --infer): ConQueR keeps every
    (query, class) with score >= 0.1; Voxel-DETR the 300 best pairs."""
    from golden_init import INFER_OVERRIDES

    fixture, yaml_name = INFER[which]
    model, g = _build(device, full_graph=False, fixture=fixture, yaml_name=yaml_name, extra=INFER_OVERRIDES)
    model.eval()
    points_list, _ = full_inputs()
    for i, pts in enumerate(points_list):
        with torch.no_grad(), install():
            res = model([({"points": torch.from_numpy(pts).to(device)}, {})])[0]
        want = {k: g["%s::%d" % (k, i)] for k in ("scores", "labels", "boxes3d")}
        got = {k: res[k].numpy() for k in want}
        assert got["scores"].shape == want["scores"].shape, (got["scores"].shape, want["scores"].shape)
        if which == "voxeldetr":
            assert got["scores"].shape[0] == 300
        # the order of the result rows follows the encoder's `topk(..., sorted=False)` proposals ($CQ/transformer.py:65),
        # which the reference leaves to the backend (its CPU and CUDA runs differ): compare as sets, ordered by score
        og, ow = np.argsort(-got["scores"], kind="stable"), np.argsort(-want["scores"], kind="stable")
        got, want = {k: v[og] for k, v in got.items()}, {k: v[ow] for k, v in want.items()}
        np.testing.assert_allclose(got["scores"], want["scores"], rtol=0, atol=2e-5)
        np.testing.assert_array_equal(got["labels"], want["labels"])
        np.testing.assert_allclose(got["boxes3d"], want["boxes3d"], rtol=0, atol=2e-4)


@pytest.mark.parametrize("which", ["conquer", "voxeldetr"])
def test_reference_inference_cpu(which, oracle_mod):
    from oracle import cpu_backend

    torch.set_num_threads(8)
    _check_inference(which, torch.device("cpu"), cpu_backend.install)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["conquer", "voxeldetr"])
def test_reference_inference_gpu(which, dev):
    import contextlib

    _check_inference(which, dev, contextlib.nullcontext)
