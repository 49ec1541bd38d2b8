"""GPU parity: box / MS-deformable attention HIP kernels vs golden vectors from the reference's
ms_deform_attn_core_pytorch and vs the oracle at ConQueR sizes.  Tolerances (fp32):
forward 1e-5, backward 1e-4."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "msda_*.npz")))


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("api", ["box", "msda"])
def test_forward_backward_golden(dev, case, api):
    from efg_amd.operators.box_attention_func import BoxAttnFunction
    from efg_amd.operators.ms_deform_attn import MSDeformAttnFunction

    g = golden(case)
    value, loc, attn = (_t(g[k], dev).requires_grad_(True) for k in ("value", "loc", "attn"))
    fn = BoxAttnFunction if api == "box" else MSDeformAttnFunction
    out = fn.apply(value, _t(g["shapes"], dev), _t(g["level_start"], dev), loc, attn, 64)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out_fp64"], rtol=1e-5, atol=1e-5)
    out.backward(_t(g["grad_out"], dev))
    np.testing.assert_allclose(value.grad.cpu().numpy(), g["grad_value"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(attn.grad.cpu().numpy(), g["grad_attn"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(loc.grad.cpu().numpy(), g["grad_loc"], rtol=1e-4, atol=2e-4)


def test_conquer_encoder_size_vs_oracle(dev, oracle_mod):
    """B=1 slice of the encoder shape: 188x188 map, 8 heads x 32, 25 points, Lq = 4000 queries."""
    from efg_amd.operators.box_attention_func import box_attn_backward, box_attn_forward

    rng = np.random.default_rng(3)
    b, hh, ww, h, d, lq, p = 1, 188, 188, 8, 32, 4000, 25
    value = rng.standard_normal((b, hh * ww, h, d), dtype=np.float32)
    loc = rng.uniform(-0.05, 1.05, (b, lq, h, 1, p, 2)).astype(np.float32)
    attn = rng.uniform(0, 1, (b, lq, h, 1, p)).astype(np.float32)
    attn /= attn.sum(-1, keepdims=True)
    go = rng.standard_normal((b, lq, h * d), dtype=np.float32)
    shapes = np.array([[hh, ww]], np.int64)
    start = np.array([0], np.int64)
    out = box_attn_forward(_t(value, dev), _t(shapes, dev), _t(start, dev), _t(loc, dev), _t(attn, dev), 64)
    np.testing.assert_allclose(out.cpu().numpy(), oracle_mod.msda_forward(value, shapes, start, loc, attn), rtol=1e-5,
                               atol=1e-5)
    gv, gl, ga = box_attn_backward(_t(value, dev), _t(shapes, dev), _t(start, dev), _t(loc, dev), _t(attn, dev),
                                   _t(go, dev), 64)
    egv, egl, ega = oracle_mod.msda_backward(value, shapes, start, loc, attn, go)
    np.testing.assert_allclose(gv.cpu().numpy(), egv, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ga.cpu().numpy(), ega, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gl.cpu().numpy(), egl, rtol=1e-4, atol=5e-4)


def test_linearity_full_size(dev):
    """Size-independent property at the full encoder size (B=2, Lq=35344): the op is linear in
    `value` and in `attn`."""
    from efg_amd.operators.box_attention_func import box_attn_forward

    g = torch.Generator(device="cpu").manual_seed(0)
    b, s, h, d, lq, p = 2, 188 * 188, 8, 32, 188 * 188, 25
    v1 = torch.randn(b, s, h, d, generator=g).to(dev)
    v2 = torch.randn(b, s, h, d, generator=g).to(dev)
    loc = torch.rand(b, lq, h, 1, p, 2, generator=g).to(dev)
    attn = torch.softmax(torch.randn(b, lq, h, p, generator=g), -1).view(b, lq, h, 1, p).to(dev)
    shapes = torch.tensor([[188, 188]], device=dev)
    start = torch.zeros(1, dtype=torch.int64, device=dev)
    o1 = box_attn_forward(v1, shapes, start, loc, attn, 64)
    o2 = box_attn_forward(v2, shapes, start, loc, attn, 64)
    o12 = box_attn_forward(v1 + 2 * v2, shapes, start, loc, attn, 64)
    torch.testing.assert_close(o12, o1 + 2 * o2, rtol=1e-4, atol=1e-4)
    o_half = box_attn_forward(v1, shapes, start, loc, attn * 0.5, 64)
    torch.testing.assert_close(o_half, o1 * 0.5, rtol=1e-5, atol=1e-6)


def test_errors(dev):
    from efg_amd.operators.box_attention_func import box_attn_forward

    shapes = torch.tensor([[4, 4]], device=dev)
    start = torch.zeros(1, dtype=torch.int64, device=dev)
    v = torch.zeros(3, 16, 2, 8, device=dev)
    loc = torch.zeros(3, 5, 2, 1, 4, 2, device=dev)
    attn = torch.zeros(3, 5, 2, 1, 4, device=dev)
    with pytest.raises(RuntimeError):  # batch 3 does not divide im2col_step 2 (box_attn.cu:39-41)
        box_attn_forward(v, shapes, start, loc, attn, 2)
    with pytest.raises(RuntimeError):  # non-contiguous input (CHECK_INPUT)
        box_attn_forward(v.transpose(2, 3).contiguous().transpose(2, 3), shapes, start, loc, attn, 64)
    with pytest.raises(RuntimeError):  # CPU tensors: no fallback
        box_attn_forward(v.cpu(), shapes.cpu(), start.cpu(), loc.cpu(), attn.cpu(), 64)
