"""GPU: fused BatchNorm1d (+ residual) (+ ReLU) (csrc/batchnorm.hip) against nn.BatchNorm1d + add + relu in fp32:
outputs / running statistics 1e-5, gradients 1e-4 relative to the tensor max."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,c", [(117697, 32), (45760, 64), (5926, 256), (700, 16), (3, 128), (1, 64), (9000, 96)])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False)])
def test_bn_act(m, c, relu, with_res):
    from efg_amd.operators.batchnorm import bn_act

    g = torch.Generator().manual_seed(m + c)
    bn = torch.nn.BatchNorm1d(c).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(c, generator=g) * 0.5 + 1)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
    ref_bn = copy.deepcopy(bn)
    x = (torch.randn(m, c, generator=g) * 1.7 + 0.4).cuda().requires_grad_(True)
    r = torch.randn(m, c, generator=g).cuda().requires_grad_(True) if with_res else None
    dy = torch.randn(m, c, generator=g).cuda()
    if m == 1:  # nn.BatchNorm1d refuses a single value per channel in training mode; the fused op must agree in spirit
        with pytest.raises(ValueError):
            ref_bn(x)
        return

    def run(fn, mod):
        for t in (x, r, mod.weight, mod.bias):
            if t is not None:
                t.grad = None
        y = fn()
        y.backward(dy)
        return y.detach(), [t.grad.clone() if t is not None else None for t in (x, r, mod.weight, mod.bias)]

    def ref_fn():
        y = ref_bn(x)
        if with_res:
            y = y + r
        return torch.relu(y) if relu else y

    y_ref, g_ref = run(ref_fn, ref_bn)
    y, gr = run(lambda: bn_act(x, bn, relu=relu, residual=r), bn)
    torch.testing.assert_close(y, y_ref, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, ref_bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    for a, b, name in zip(gr, g_ref, ("x", "residual", "weight", "bias")):
        if b is None:
            assert a is None
            continue
        scale = float(b.abs().max()) + 1e-12
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)


def test_eval_mode_uses_the_module():
    from efg_amd.operators.batchnorm import bn_act

    bn = torch.nn.BatchNorm1d(32).cuda().eval()
    x = torch.randn(100, 32, device="cuda")
    torch.testing.assert_close(bn_act(x, bn, relu=True), torch.relu(bn(x)))
