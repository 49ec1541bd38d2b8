"""GPU: the fused Box3dAttention path (csrc/box_fused.hip: geometry + softmax + sampling in one kernel)
against the reference sequence (_where_to_attend + softmax + BoxAttnFunction, $CQ/modules/box_attention.py:62-115)
evaluated with the golden-pinned unfused op.  Forward 1e-5, gradients 1e-4 relative to the tensor max."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _module(dev, rot, seed):
    from efg_amd.detection3d.box_attention import Box3dAttention

    torch.manual_seed(seed)
    m = Box3dAttention(256, 1, 8, with_rotation=rot).to(dev)
    with torch.no_grad():  # the reference initialises these to zero; make them matter
        m.linear_box_weight.normal_(0, 0.02)
        m.linear_attn_weight.normal_(0, 0.05)
    return m


def _run(m, fused, query, value, shapes, start, ref):
    import efg_amd.operators.box_attention_func as baf

    old = baf.FUSED_ENABLED
    baf.FUSED_ENABLED = fused
    try:
        q = query.clone().requires_grad_(True)
        v = value.clone().requires_grad_(True)
        m.zero_grad()
        out = m(q, v, shapes, None, start, None, ref)[0]
        g = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).to(out.device)
        out.backward(g)
        return out.detach(), q.grad, v.grad, {n: p.grad.clone() for n, p in m.named_parameters()}
    finally:
        baf.FUSED_ENABLED = old


@pytest.mark.parametrize("case", ["encoder_grid", "encoder_grid_rot", "encoder_grid_big", "decoder_rot",
                                  "decoder_rot_many", "decoder_norot_small"])
def test_fused_matches_reference_sequence(dev, case):
    g = torch.Generator().manual_seed(1)
    H, W = 40, 36
    S = H * W
    shapes = torch.tensor([[H, W]], device=dev)
    start = torch.zeros(1, dtype=torch.int64, device=dev)
    value = torch.randn(2, S, 256, generator=g).to(dev)
    if case.startswith("encoder_grid"):   # queries are the map cells -> tile (GEMM-form) backward
        rot, lq = case.endswith("rot"), S
        ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
        ref = torch.zeros(2, S, 7)
        ref[..., 0], ref[..., 1] = (xs / W).reshape(-1), (ys / H).reshape(-1)
        ref[..., 2] = ref[..., 5] = 0.5
        ref[..., 3] = ref[..., 4] = 0.08
        if case.endswith("big"):   # 14-cell boxes: most sampling points leave the 16x16 window (global path)
            ref[..., 3] = ref[..., 4] = 0.4
        if rot:
            ref[..., 6] = torch.rand(2, S, generator=g)
    else:
        rot, lq = case.startswith("decoder_rot"), (400 if case.endswith("many") else 77)  # many: binned grad_value
        ref = torch.rand(2, lq, 7, generator=g)
        ref[..., 3:5] = ref[..., 3:5] * 0.2 + 0.02
        if case == "decoder_norot_small":
            ref[..., 3] *= -1.0   # negative width -> relu gate closes (gradient mask path)
    ref = ref.to(dev)
    query = torch.randn(2, lq, 256, generator=g).to(dev)
    m = _module(dev, rot, 3)
    o1, q1, v1, p1 = _run(m, False, query, value, shapes, start, ref)
    o2, q2, v2, p2 = _run(m, True, query, value, shapes, start, ref)
    torch.testing.assert_close(o2, o1, rtol=1e-5, atol=1e-5)

    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-12
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)

    close(v2, v1, "grad value")
    close(q2, q1, "grad query")
    for n in p1:
        close(p2[n], p1[n], n)
