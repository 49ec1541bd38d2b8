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


# ---- fused HIP kernels directly against the CPU oracle (no HIP op on the expected side) ------------------------------
class _OracleSampling(torch.autograd.Function):
    """oracle.msda_forward / msda_backward (the C restatement of box_attn_kernel.cuh:274-472, pinned to
    ms_deform_attn_core_pytorch by tests/test_oracle_msda.py) as a CPU autograd node, so that torch's CPU autograd carries
    the gradient on through the reference's geometry + softmax ($CQ/modules/box_attention.py:62-115)."""

    @staticmethod
    def forward(ctx, value, shapes, start, loc, attn):
        import oracle

        ctx.save_for_backward(value, shapes, start, loc, attn)
        return torch.from_numpy(oracle.msda_forward(value.numpy(), shapes.numpy(), start.numpy(), loc.numpy(), attn.numpy()))

    @staticmethod
    def backward(ctx, grad_out):
        import oracle

        value, shapes, start, loc, attn = ctx.saved_tensors
        gv, gl, ga = oracle.msda_backward(value.numpy(), shapes.numpy(), start.numpy(), loc.numpy(), attn.numpy(),
                                          grad_out.contiguous().numpy())
        return torch.from_numpy(gv), None, None, torch.from_numpy(gl), torch.from_numpy(ga)


def _oracle_case(case, g):
    H, W, heads, P = 40, 36, 8, 25
    S = H * W
    rot = case in ("decoder_rot_binned", "encoder_mixed_rot")
    nvar = 5 if rot else 4
    if case.startswith("encoder"):
        lq = S
        ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
        ref = torch.zeros(2, S, 7)
        ref[..., 0], ref[..., 1] = (xs / W).reshape(-1), (ys / H).reshape(-1)
        ref[..., 2] = ref[..., 5] = 0.5
        if case == "encoder_big":      # 14-cell boxes: nearly every corner leaves the 16x16 tile window (binned path)
            ref[..., 3] = ref[..., 4] = 0.4
        else:                          # a mix: most queries stay inside the window, the larger ones straddle it
            size = torch.rand(2, S, generator=g) ** 2 * 0.5 + 0.04
            ref[..., 3] = size
            ref[..., 4] = size * (0.7 + 0.6 * torch.rand(2, S, generator=g))
        if rot:
            ref[..., 6] = torch.rand(2, S, generator=g)
    else:
        lq = 400                       # 2 * 400 * 8 * 25 * 4 corners > kBinMinEntries: the decoder's binned path
        ref = torch.rand(2, lq, 7, generator=g)
        ref[..., 3:5] = ref[..., 3:5] * 0.2 + 0.02
    value = torch.randn(2, S, heads, 32, generator=g)
    offsets = torch.randn(2, lq, heads * nvar, generator=g) * 0.5
    logits = torch.randn(2, lq, heads * P, generator=g)
    idx = torch.linspace(-2, 2, 5) / 5.0   # any fixed 5 x 5 lattice inside (-0.5, 0.5) will do for this comparison
    ky, kx = torch.meshgrid(idx, idx, indexing="ij")
    kidx = torch.stack((kx, ky), -1).reshape(-1, 2).contiguous()
    return value, ref, offsets, logits, kidx, (H, W, heads, P, rot, nvar, lq)


@pytest.mark.parametrize("case", ["encoder_big", "encoder_mixed", "encoder_mixed_rot", "decoder_rot_binned"])
def test_fused_kernels_match_oracle(dev, oracle_mod, case):
    """out, grad_value, grad_offsets and grad_logits of the fused kernels (tile backward + the binned out-of-window
    path, decoder binned path) against the CPU oracle's sampling op chained with the reference geometry in torch-CPU
    autograd.  The mixed cases assert that a real share of the corners leaves the tile window."""
    import efg_amd.operators.box_attention_func as baf

    g = torch.Generator().manual_seed(11)
    value, ref, offsets, logits, kidx, (H, W, heads, P, rot, nvar, lq) = _oracle_case(case, g)
    shapes, start = torch.tensor([[H, W]]), torch.zeros(1, dtype=torch.int64)
    gout = torch.randn(2, lq, heads * 32, generator=g)

    # expected: CPU, oracle sampling
    v_c, o_c, l_c = (t.clone().requires_grad_(True) for t in (value, offsets, logits))
    grid = baf.box_sampling_grid(ref, o_c, kidx, heads, 1, rot)
    attn = torch.softmax(l_c.view(2, lq, heads, P), dim=-1).view(2, lq, heads, 1, P)
    exp = _OracleSampling.apply(v_c, shapes, start, grid, attn)
    exp.backward(gout)

    if case.startswith("encoder_mixed"):   # how many corners fall outside the window of their 4 x 8 query tile
        tqy = 4      # csrc/box_fused.hip: BT<4>, window = tile + 4 cells
        px = grid.detach()[..., 0] * W - 0.5
        py = grid.detach()[..., 1] * H - 0.5
        q = torch.arange(lq)
        wx0 = ((q % W) // 8 * 8 - 4).view(1, lq, 1, 1, 1)
        wy0 = ((q // W) // tqy * tqy - 4).view(1, lq, 1, 1, 1)
        out_of_win = ((torch.floor(px) < wx0) | (torch.floor(px) + 1 > wx0 + 15) | (torch.floor(py) < wy0)
                      | (torch.floor(py) + 1 > wy0 + tqy + 7))
        frac = float(out_of_win.float().mean())
        assert 0.03 < frac < 0.5, "the case should mix in-window and out-of-window corners, got %.3f" % frac

    # actual: the fused HIP op
    v_d, o_d, l_d = (t.to(dev).requires_grad_(True) for t in (value, offsets, logits))
    out = baf.BoxAttnFusedFunction.apply(v_d, shapes.to(dev), start.to(dev), ref.to(dev), o_d, l_d, kidx.to(dev), nvar)
    out.backward(gout.to(dev))

    def close(a, b, name, tol):
        scale = float(b.abs().max()) + 1e-12
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=tol, atol=tol * scale, err_msg=name)

    close(out, exp, "out", 1e-5)
    close(v_d.grad, v_c.grad, "grad_value", 1e-4)
    close(o_d.grad, o_c.grad, "grad_offsets", 1e-4)
    close(l_d.grad, l_c.grad, "grad_logits", 1e-4)



@pytest.mark.parametrize("case", ["encoder_grid", "encoder_grid_big", "decoder_many"])
def test_backward_is_reproducible_bit_for_bit(dev, case):
    """Two identical backward passes give identical bits in every gradient.  Encoder: the query tiles run in colour
    classes whose windows never overlap (plain read-modify-write flush, one launch per class); out-of-window and decoder
    corners are binned and each bin summed as exact fixed-point integers, whatever order its entries arrived in
    (csrc/box_fused.hip).  `encoder_grid_big`: most corners leave the window; `decoder_many`: 2 x 2000 free queries."""
    g = torch.Generator().manual_seed(2)
    H, W = 52, 44
    S = H * W
    shapes = torch.tensor([[H, W]], device=dev)
    start = torch.zeros(1, dtype=torch.int64, device=dev)
    value = torch.randn(2, S, 256, generator=g).to(dev)
    if case.startswith("encoder"):
        lq = S
        ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
        ref = torch.zeros(2, S, 7)
        ref[..., 0], ref[..., 1] = (xs / W).reshape(-1), (ys / H).reshape(-1)
        ref[..., 2] = ref[..., 5] = 0.5
        ref[..., 3] = ref[..., 4] = 0.4 if case.endswith("big") else 0.08
        rot = False
    else:
        lq, rot = 2000, True
        ref = torch.rand(2, lq, 7, generator=g)
        ref[..., 3:5] = ref[..., 3:5] * 0.2 + 0.02
    ref = ref.to(dev)
    query = torch.randn(2, lq, 256, generator=g).to(dev)
    m = _module(dev, rot, 4)
    first = _run(m, True, query, value, shapes, start, ref)
    for trial in range(3):
        again = _run(m, True, query, value, shapes, start, ref)
        assert torch.equal(again[2], first[2]), "grad_value differs between two identical runs (%s)" % case
        assert torch.equal(again[1], first[1]) and torch.equal(again[0], first[0])
        for n in first[3]:
            assert torch.equal(again[3][n], first[3][n]), n


@pytest.mark.parametrize("case", ["grid", "grid_rot", "grid_big"])
def test_split_backward_bits_equal_one_kernel(dev, monkeypatch, case):
    """The encoder backward as two kernels at four waves per SIMD (box_bwd_tile_a_kernel: G = GO.V^T + the gradients of
    offsets and logits; box_bwd_tile_b_kernel: W scatter + W.GO + flush; EFG_BOX_SPLIT=1, the default) performs the one-kernel
    form's arithmetic in its order: every gradient is the same bits (near boxes, rotated boxes, boxes whose corners leave the
    window and take the binned path)."""
    g = torch.Generator().manual_seed(4)
    H, W = 44, 52
    S = H * W
    shapes = torch.tensor([[H, W]], device=dev)
    start = torch.zeros(1, dtype=torch.int64, device=dev)
    value = torch.randn(2, S, 256, generator=g).to(dev)
    rot = case.endswith("rot")
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref = torch.zeros(2, S, 7)
    ref[..., 0], ref[..., 1] = (xs / W).reshape(-1), (ys / H).reshape(-1)
    ref[..., 2] = ref[..., 5] = 0.5
    ref[..., 3] = ref[..., 4] = 0.4 if case.endswith("big") else 0.08
    if rot:
        ref[..., 6] = torch.rand(2, S, generator=g)
    ref = ref.to(dev)
    query = torch.randn(2, S, 256, generator=g).to(dev)
    m = _module(dev, rot, 5)
    res = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("EFG_BOX_SPLIT", sw)
        res[sw] = _run(m, True, query, value, shapes, start, ref)
    o1, q1, v1, p1 = res["1"]
    o0, q0, v0, p0 = res["0"]
    assert torch.equal(o1, o0) and torch.equal(v1, v0), "grad_value bits differ"
    assert torch.equal(q1, q0), "query gradient (through grad_offsets / grad_logits) bits differ"
    for n in p0:
        assert torch.equal(p1[n], p0[n]), n
