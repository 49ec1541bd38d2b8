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


def test_channels_last_conv_stack_matches_the_modules():
    """operators/batchnorm.py `run_sequential`: BatchNorm2d + ReLU over a channels-last map through the row kernels vs the
    nn.Sequential it stands in for, evaluated in fp64 -- outputs, input / parameter gradients, running statistics.
    (The fp32 modules themselves are the weaker yardstick here: MIOpen's channels-last batch-norm backward is 3-4e-3
    off the fp64 gradients on this stack, the row kernels 5e-7.)"""
    from torch import nn

    from efg_amd.operators.batchnorm import fusable_nhwc, run_sequential

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    seq = nn.Sequential(nn.ZeroPad2d(1), nn.Conv2d(32, 64, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                        nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                        nn.Conv2d(64, 8, 3, padding=1)).to(dev).to(memory_format=torch.channels_last)
    ref = copy.deepcopy(seq).double()
    x = torch.randn(2, 32, 47, 53, device=dev).contiguous(memory_format=torch.channels_last)
    assert fusable_nhwc(seq[2], seq[1](seq[0](x)))
    xa, xb = x.clone().requires_grad_(True), x.double().requires_grad_(True)
    ya, yb = run_sequential(seq, xa), ref(xb)
    assert ya.is_contiguous(memory_format=torch.channels_last)
    w = torch.randn_like(ya)
    (ya * w).sum().backward()
    (yb * w.double()).sum().backward()

    def close(a, b, tol):
        err = float((a.double() - b).norm() / b.norm().clamp_min(1e-30))
        assert err <= tol, err

    close(ya, yb, 2e-6)
    close(xa.grad, xb.grad, 5e-6)
    for pa, pb in zip(seq.parameters(), ref.parameters()):
        close(pa.grad, pb.grad, 5e-6)
    for (name, ba), bb in zip(seq.named_buffers(), ref.buffers()):
        if "num_batches" in name:
            assert int(ba) == int(bb) == 1
        else:
            close(ba, bb, 2e-6)
    seq.eval(), ref.eval()
    close(run_sequential(seq, x), ref(x.double()), 2e-6)    # eval mode: the modules themselves
