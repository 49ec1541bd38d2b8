"""GPU: fused residual-add + LayerNorm (csrc/layernorm.hip) against torch's fp32 add + layer_norm.
Forward 1e-5, gradients 1e-4 relative to the tensor max (different but equally valid fp32 summation orders)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 5000, 256), (3, 77, 256), (1, 1, 256), (700, 64), (33, 512), (9, 1024),
                                   (2, 0, 256)])
@pytest.mark.parametrize("with_residual", [True, False])
def test_add_layer_norm(shape, with_residual):
    from efg_amd.operators.layernorm import add_layer_norm

    g = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    norm = torch.nn.LayerNorm(c).cuda()
    with torch.no_grad():
        norm.weight.copy_(torch.randn(c, generator=g) * 0.5 + 1)
        norm.bias.copy_(torch.randn(c, generator=g))
    x = (torch.randn(shape, generator=g) * 2 + 0.3).cuda().requires_grad_(True)
    r = torch.randn(shape, generator=g).cuda().requires_grad_(True) if with_residual else None
    dy = torch.randn(shape, generator=g).cuda()

    def run(fn):
        for t in (x, r, norm.weight, norm.bias):
            if t is not None:
                t.grad = None
        y = fn()
        y.backward(dy)
        return y.detach(), [t.grad.clone() if t is not None else None for t in (x, r, norm.weight, norm.bias)]

    y_ref, g_ref = run(lambda: norm(x + r if with_residual else x))
    y, gr = run(lambda: add_layer_norm(x, r, norm))
    torch.testing.assert_close(y, y_ref, rtol=1e-5, atol=1e-5)
    for a, b, name in zip(gr, g_ref, ("x", "residual", "weight", "bias")):
        if b is None:
            assert a is None
            continue
        scale = float(b.abs().max()) + 1e-12 if b.numel() else 1.0
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)
