"""Channels-last GroupNorm (csrc/batchnorm.hip efg_gn_*, operators/groupnorm.py) against torch.nn.functional.group_norm."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [((2, 188, 188, 256), 32), ((1, 50, 7, 64), 8), ((3, 1000, 32), 4), ((2, 33, 16), 4), ((4, 5, 3, 1024), 32),
         ((2, 1, 1, 8), 2)]


@pytest.mark.parametrize("shape,groups", CASES)
def test_group_norm_nhwc_matches_torch(shape, groups):
    from efg_amd.operators.groupnorm import group_norm_nhwc

    torch.manual_seed(sum(shape) + groups)
    c = shape[-1]
    gn = torch.nn.GroupNorm(groups, c).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(*shape, device="cuda") * 2 + 0.7)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    y = group_norm_nhwc(xa, gn)
    to_nchw = lambda t: t.movedim(-1, 1)  # noqa: E731
    yr = F.group_norm(to_nchw(xb), groups, gn.weight, gn.bias, gn.eps).movedim(1, -1)
    assert y.shape == x.shape and y.is_contiguous()
    assert torch.allclose(y, yr, rtol=1e-5, atol=2e-5), float((y - yr).abs().max())
    up = torch.randn_like(y)
    gw, gb = torch.autograd.grad((y * up).sum(), [gn.weight, gn.bias], retain_graph=True)
    (y * up).sum().backward(inputs=[xa])
    grw, grb = torch.autograd.grad((yr * up).sum(), [gn.weight, gn.bias], retain_graph=True)
    (yr * up).sum().backward(inputs=[xb])
    n = x.numel() // c
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=2e-5), float((xa.grad - xb.grad).abs().max())
    assert torch.allclose(gw, grw, rtol=1e-4, atol=3e-6 * n ** 0.5 * 4 + 1e-5)
    assert torch.allclose(gb, grb, rtol=1e-4, atol=3e-6 * n ** 0.5 * 4 + 1e-5)
    # deterministic
    y2 = group_norm_nhwc(x.clone().requires_grad_(True), gn)
    assert torch.equal(y, y2)


def test_group_norm_nhwc_statistics_fp64():
    """Normalised output has zero mean / unit variance per (sample, group) to fp32 accuracy (weight 1, bias 0)."""
    from efg_amd.operators.groupnorm import group_norm_nhwc

    gn = torch.nn.GroupNorm(32, 256).cuda()
    x = torch.randn(2, 35344, 256, device="cuda") * 5 + 100.0  # large offset: a one-pass E[x^2] - E[x]^2 would fail
    y = group_norm_nhwc(x, gn).double().view(2, 35344, 32, 8)
    assert float(y.mean(dim=(1, 3)).abs().max()) < 1e-4
    assert float((y.var(dim=(1, 3), unbiased=False) - 1).abs().max()) < 1e-3


def test_group_norm_nhwc_rejects_unsupported():
    from efg_amd.operators.groupnorm import group_norm_nhwc

    gn = torch.nn.GroupNorm(3, 6).cuda()
    with pytest.raises(RuntimeError):
        group_norm_nhwc(torch.randn(2, 5, 6, device="cuda"), gn)  # 2 channels per group
    with pytest.raises(RuntimeError):
        group_norm_nhwc(torch.randn(2, 5, 6), torch.nn.GroupNorm(1, 6))  # host tensor: no CPU fallback


def test_input_projection_matches_module():
    """VoxelDETR._project (GEMM + channels-last GroupNorm) against the plain nn.Sequential on the same parameters."""
    from efg_amd.detection3d.voxel_detr import VoxelDETR
    from efg_amd.modeling.common import Conv2d

    torch.manual_seed(0)
    proj = torch.nn.Sequential(Conv2d(384, 256, kernel_size=1), torch.nn.GroupNorm(32, 256)).cuda()
    x = torch.randn(2, 384, 94, 94, device="cuda").contiguous(memory_format=torch.channels_last)
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = VoxelDETR._project(proj, a)
    ref = torch.nn.Sequential(torch.nn.Conv2d(384, 256, kernel_size=1), torch.nn.GroupNorm(32, 256)).cuda()
    ref.load_state_dict(proj.state_dict())
    yb = ref(b)
    assert ya.shape == yb.shape
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
    assert ya.permute(0, 2, 3, 1).is_contiguous()  # token layout, no copy needed downstream
    up = torch.randn_like(yb)
    (ya * up).sum().backward()
    (yb * up).sum().backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-3, atol=1e-4)
    for (n1, p1), (n2, p2) in zip(proj.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p1.grad, p2.grad, rtol=1e-3, atol=2e-3), n1
