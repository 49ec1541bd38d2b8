"""GPU: the two sparse backbones of the reference (ConQueR SparseResNet res18 and CenterPoint
SpMiddleResNetFHD, efg/modeling/backbones/sparse_net.py:239-316,473-545) on the HIP ops against the same
modules on the CPU with oracle ops: output shapes / channel plans of the reference, values 1e-4, input grads."""
import contextlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RES18 = dict(depth=18, out_features=["res2", "res3", "res4"], num_groups=1, norm="BN1d",
             activation=dict(type="ReLU", inplace=True), width_per_group=64, res1_out_channels=64,
             stem_out_channels=32)


def _inputs(nf, seed=0):
    from efg_amd.data.synthetic import make_scene

    feats, coors = [], []
    import oracle

    for b in range(2):
        pts, _, _ = make_scene(700 + b + seed, n_points=60000, n_sweeps=4 if nf == 6 else 1)
        keep = (np.abs(pts[:, 0]) < 12.8) & (np.abs(pts[:, 1]) < 12.8)
        pts = pts[keep][:5000]
        v, c, n = oracle.hard_voxelize(pts, (0.1, 0.1, 0.15), (-12.8, -12.8, -2.0, 12.8, 12.8, 4.0), 5, 20000)
        feats.append(oracle.voxel_mean(v, n))
        coors.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    return np.concatenate(feats), np.concatenate(coors)


def _run(build, device, ctx, feats, coors):
    torch.manual_seed(0)
    net = build().to(device)
    net.train()
    f = torch.from_numpy(feats).to(device).requires_grad_(True)
    c = torch.from_numpy(coors).to(device)
    with ctx:
        out = net(f, c, 2, [256, 256, 40])
        outs = out if isinstance(out, dict) else {"out": out}
        loss = sum((o * torch.linspace(0.5, 1.5, o.numel(), device=o.device).view_as(o)).sum() for o in outs.values())
        loss.backward()
    return {k: v.detach().cpu() for k, v in outs.items()}, f.grad.cpu()


@pytest.mark.parametrize("which", ["sparse_resnet18", "centerpoint_fhd"])
def test_backbone_matches_cpu_oracle(dev, oracle_mod, which):
    from oracle import cpu_backend

    from efg_amd.modeling.backbones import SpMiddleResNetFHD, build_sparse_resnet_backbone

    torch.set_num_threads(8)
    if which == "sparse_resnet18":
        nf, build = 5, (lambda: build_sparse_resnet_backbone(RES18, 5))
        shapes = {"res2": (2, 384, 64, 64), "res3": (2, 384, 32, 32), "res4": (2, 512, 16, 16)}  # C*D with D = 6,3,2
    else:
        nf, build = 6, (lambda: SpMiddleResNetFHD(num_input_features=6))
        shapes = {"out": (2, 256, 32, 32)}                                                      # 128 x D=2
    feats, coors = _inputs(nf)
    o_cpu, g_cpu = _run(build, torch.device("cpu"), cpu_backend.install(), feats, coors)
    o_gpu, g_gpu = _run(build, dev, contextlib.nullcontext(), feats, coors)
    for k, shp in shapes.items():
        assert tuple(o_gpu[k].shape) == shp
        np.testing.assert_allclose(o_gpu[k].numpy(), o_cpu[k].numpy(), atol=1e-4 * float(o_cpu[k].abs().max()), rtol=1e-4)
    np.testing.assert_allclose(g_gpu.numpy(), g_cpu.numpy(), atol=2e-3 * float(g_cpu.abs().max()), rtol=2e-3)
