"""The single-node halves of the encoder layer (efg_amd/detection3d/encoder_layer.py) against the module-by-module form of
the same layer ($CQ/transformer.py:206-243): the same forward kernels -> identical outputs; the input gradient is the
same sum accumulated in another order (GEMM beta = 1 instead of addition kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_fused_halves_equal_module_form(dev):
    from efg_amd.detection3d import encoder_layer as enc
    from efg_amd.detection3d.transformer import TransformerEncoderLayer

    torch.manual_seed(0)
    b, hh, ww, c = 2, 96, 96, 256          # 18 432 tokens: above the long-matrix threshold of operators/linear.py
    layer = TransformerEncoderLayer(c, 8, 1, 1024, 0.0).to(dev)
    with torch.no_grad():   # non-trivial attention: the module starts with zero box / attention weights
        layer.self_attn.linear_attn_weight.normal_(0, 0.02)
        layer.self_attn.linear_box_weight.normal_(0, 0.02)
    n = hh * ww
    src0 = torch.randn(b, n, c, device=dev)
    pos = torch.randn(b, n, c, device=dev) * 0.1
    ys, xs = torch.meshgrid(torch.arange(hh, device=dev), torch.arange(ww, device=dev), indexing="ij")
    ref = torch.zeros(b, n, 7, device=dev)
    ref[..., 0] = ((xs.reshape(-1) + 0.5) / ww)[None]
    ref[..., 1] = ((ys.reshape(-1) + 0.5) / hh)[None]
    ref[..., 3] = 4.0 / ww
    ref[..., 4] = 4.0 / hh
    shape = torch.tensor([[hh, ww]], device=dev, dtype=torch.int64)
    start = torch.zeros(1, device=dev, dtype=torch.int64)
    go = torch.randn(b, n, c, device=dev)

    def run(fused):
        saved = enc._ENABLED
        enc._ENABLED = fused
        try:
            layer.zero_grad(set_to_none=True)
            src = src0.clone().requires_grad_(True)
            assert enc.usable(layer, src, pos, ref) == fused
            y = layer(src, pos, shape, start, ref)
            y.backward(go)
            return y.detach().cpu().numpy(), src.grad.cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in layer.named_parameters()}
        finally:
            enc._ENABLED = saved

    y0, gx0, gp0 = run(False)
    y1, gx1, gp1 = run(True)
    assert np.array_equal(y0, y1), "the forward kernels are the same: outputs must be identical"
    scale = np.abs(gx0).max()
    assert np.abs(gx1 - gx0).max() <= 2e-6 * scale, (np.abs(gx1 - gx0).max(), scale)
    assert set(gp0) == set(gp1)
    for k in gp0:
        s = np.abs(gp0[k]).max() + 1e-30
        assert np.abs(gp1[k] - gp0[k]).max() <= 2e-5 * s, (k, np.abs(gp1[k] - gp0[k]).max(), s)
