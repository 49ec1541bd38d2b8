"""csrc/colsum.hip (the bias gradient of the path's Linear layers) and operators/linear.py against PyTorch."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(70688, 256), (70688, 1024), (70688, 200), (70688, 32), (2480, 256), (2800, 256), (2480, 3), (2480, 7),
          (2480, 1), (5, 10), (1, 256), (17, 1028), (63, 260), (16, 4), (100000, 8), (33, 65)]


def _ref(x):
    return x.double().sum(0)


@pytest.mark.parametrize("rows,cols", SHAPES)
def test_column_sum_matches_fp64(rows, cols):
    from efg_amd.operators.linear import column_sum

    g = torch.Generator(device="cuda").manual_seed(rows * 7 + cols)
    x = torch.randn(rows, cols, device="cuda", generator=g) * 3 + 0.25
    out = column_sum(x)
    ref = _ref(x)
    # fp32 tree sum: a few ulps of the partial sums, whose size is ~ |mean| * rows + 3 * sqrt(rows)
    tol = 4e-7 * (0.25 * rows + 3 * rows ** 0.5) * 4 + 1e-6
    assert out.shape == (cols,)
    assert float((out.double() - ref).abs().max()) <= tol, (rows, cols, float((out.double() - ref).abs().max()), tol)
    # and at least as accurate as ATen's own fp32 reduction (x4 slack)
    aten = float((x.sum(0).double() - ref).abs().max())
    assert float((out.double() - ref).abs().max()) <= 4 * aten + 1e-6
    assert torch.equal(out, column_sum(x))  # deterministic


def test_column_sum_views():
    from efg_amd.operators.linear import column_sum

    g = torch.Generator(device="cuda").manual_seed(3)
    wide = torch.randn(5000, 256, device="cuda", generator=g)
    for view in (wide[:, :200], wide[:, 1:], wide[:, 4:132], wide[7:, 3:6], wide.t()[:, :77]):
        out = column_sum(view)
        assert float((out.double() - _ref(view)).abs().max()) < 2e-3
    empty = torch.empty(0, 256, device="cuda")
    assert torch.equal(column_sum(empty), torch.zeros(256, device="cuda"))


def test_column_sum_exact_on_integers():
    """Small integers sum exactly in fp32 whatever the order: the reduction must be bit-exact."""
    from efg_amd.operators.linear import column_sum

    g = torch.Generator(device="cuda").manual_seed(5)
    for rows, cols in [(70688, 256), (2480, 200), (999, 33)]:
        x = torch.randint(-8, 9, (rows, cols), device="cuda", generator=g).float()
        assert torch.equal(column_sum(x), x.double().sum(0).float())


def test_column_sum_single_launch_tickets_wrap_and_reset():
    """Up to 128 row blocks the final sum runs inside the partial kernel, the block that draws the last ticket of a ring of
    self-resetting counters doing it: ~10 000 back-to-back calls (the ring has 8192 slots; the wide matrix takes 4 per
    call) on two streams keep giving the first call's bits."""
    from efg_amd.operators.linear import column_sum, relu_backward_column_sum

    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(2480, 256, device="cuda", generator=g)
    b = torch.randn(2800, 1024, device="cuda", generator=g)
    y = torch.relu(torch.randn(2800, 1024, device="cuda", generator=g))
    ra, rb = column_sum(a), relu_backward_column_sum(b, y)[1]
    assert torch.equal(rb, column_sum(torch.ops.aten.threshold_backward(b, y, 0)))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    bad_side = torch.zeros((), dtype=torch.int64, device="cuda")
    for i in range(2500):
        bad += (column_sum(a) != ra).sum()
        with torch.cuda.stream(side):
            bad_side += (relu_backward_column_sum(b, y)[1] != rb).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0 and int(bad_side) == 0


@pytest.mark.parametrize("shape,fin,fout", [((2, 1240, 256), 256, 256), ((1240, 2, 256), 256, 1024),
                                             ((2, 35344, 256), 256, 200), ((2, 300, 10), 10, 256), ((77, 256), 256, 3)])
def test_linear_matches_nn_linear(shape, fin, fout):
    from efg_amd.operators.linear import Linear, LinearFunction

    torch.manual_seed(0)
    ref = torch.nn.Linear(fin, fout).cuda()
    mine = Linear(fin, fout).cuda()
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(*shape, device="cuda")
    if len(shape) == 3 and shape[0] > shape[1]:
        x = x.transpose(0, 1).contiguous().transpose(0, 1)  # a strided [Q, B, C] view, as in the decoder
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    # linear() routes short matrices to F.linear; exercise the custom backward at every size
    ya, yb = ref(xa), LinearFunction.apply(xb, mine.weight, mine.bias, False)
    assert torch.equal(ya, yb)  # same addmm
    w = torch.randn_like(ya)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)
    # same product (x^T @ grad); over 70 688 rows the library's split-K order may differ between two calls
    k = x.numel() // fin
    assert torch.allclose(ref.weight.grad, mine.weight.grad, rtol=1e-4, atol=2e-6 * k ** 0.5 * float(ref.weight.grad.abs().max()) + 1e-3)
    scale = float(ref.bias.grad.abs().max()) + 1.0
    assert float((ref.bias.grad - mine.bias.grad).abs().max()) <= 2e-5 * scale * (x.numel() / fin) ** 0.5
    # inference / no-grad path is plain F.linear
    with torch.no_grad():
        assert torch.equal(mine(x), ref(x))


def test_weight_grad_split_matches_single_product():
    """The 16-chunk batched weight gradient of the 70 688-row layers against the single product, fp64 as referee."""
    from efg_amd.operators.linear import weight_grad

    g = torch.Generator(device="cuda").manual_seed(11)
    for k, cin, cout in [(70688, 256, 256), (70688, 256, 1024), (70688, 1024, 256), (70688, 256, 200), (35344, 384, 256)]:
        x = torch.randn(k, cin, device="cuda", generator=g)
        go = torch.randn(k, cout, device="cuda", generator=g)
        got = weight_grad(x, go)
        assert got.shape == (cout, cin) and got.is_contiguous()
        ref = go.double().t().mm(x.double())
        one = x.t().mm(go).t()
        scale = float(ref.abs().max())
        e_split, e_one = float((got.double() - ref).abs().max()) / scale, float((one.double() - ref).abs().max()) / scale
        assert e_split < 2e-5 and e_split <= 4 * e_one + 1e-7, (k, cin, cout, e_split, e_one)
        assert torch.equal(got, weight_grad(x, go))  # deterministic
    # short matrices keep the single product, in the parameter's own [out, in] row-major layout (AccumulateGrad then takes the
    # tensor as it is: no copy launch per parameter) -- the product autograd issues for F.linear
    x, go = torch.randn(2480, 256, device="cuda"), torch.randn(2480, 1024, device="cuda")
    got = weight_grad(x, go)
    assert got.shape == (1024, 256) and got.is_contiguous() and torch.equal(got, go.t().mm(x))


def test_linear_without_bias_and_pointwise_conv():
    from efg_amd.modeling.common import Conv2d
    from efg_amd.operators.linear import LinearFunction, linear

    torch.manual_seed(1)
    w = torch.randn(64, 32, device="cuda", requires_grad=True)
    x = torch.randn(3, 50, 32, device="cuda", requires_grad=True)
    y = LinearFunction.apply(x, w, None, False)
    y2 = torch.nn.functional.linear(x.detach(), w.detach())
    assert torch.equal(y, y2)
    y.square().sum().backward()
    wr, xr = w.detach().clone().requires_grad_(True), x.detach().clone().requires_grad_(True)
    torch.nn.functional.linear(xr, wr).square().sum().backward()
    assert torch.allclose(w.grad, wr.grad, rtol=1e-5, atol=1e-4) and torch.allclose(x.grad, xr.grad, rtol=1e-5, atol=1e-5)
    conv = Conv2d(32, 48, kernel_size=1).cuda()
    ref = torch.nn.Conv2d(32, 48, kernel_size=1).cuda()
    ref.load_state_dict(conv.state_dict())
    assert linear(x.detach().requires_grad_(True), w).grad_fn.__class__.__name__ != "LinearFunctionBackward"  # short
    xi = torch.randn(2, 32, 100, 96, device="cuda").contiguous(memory_format=torch.channels_last)  # 19 200 rows
    a, b = xi.clone().requires_grad_(True), xi.clone().requires_grad_(True)
    ya, yb = conv(a), ref(b)
    assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-5)
    ya.square().sum().backward()
    yb.square().sum().backward()
    assert torch.allclose(conv.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(conv.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-4)


def test_linear_relu_epilogue():
    """linear(..., relu=True): bias + ReLU in the GEMM epilogue, identical values, autograd-identical gradients."""
    from efg_amd.operators.linear import linear

    torch.manual_seed(2)
    w = (torch.randn(1024, 256, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(1024, device="cuda").requires_grad_(True)
    x = torch.randn(2, 10000, 256, device="cuda", requires_grad=True)
    y = linear(x, w, b, relu=True)
    assert y.grad_fn.__class__.__name__ == "LinearFunctionBackward"
    wr, br, xr = (t.detach().clone().requires_grad_(True) for t in (w, b, x))
    yr = torch.relu(torch.nn.functional.linear(xr, wr, br))
    assert torch.equal(y, yr)
    up = torch.randn_like(y)
    (y * up).sum().backward()
    (yr * up).sum().backward()
    assert torch.allclose(x.grad, xr.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(w.grad, wr.grad, rtol=1e-4, atol=2e-2)
    assert torch.allclose(b.grad, br.grad, rtol=1e-4, atol=2e-2)
    # short input: plain PyTorch
    xs = torch.randn(2, 100, 256, device="cuda", requires_grad=True)
    assert torch.equal(linear(xs, w, b, relu=True), torch.relu(torch.nn.functional.linear(xs, w, b)))


@pytest.mark.parametrize("rows,cols", [(70688, 1024), (157696, 256), (33, 64), (1, 4), (5000, 200)])
def test_relu_backward_column_sum(rows, cols):
    """efg_relu_bwd_colsum_f32: the masked gradient is bit-identical to threshold_backward, its column sums to the
    column-sum op on that masked gradient (same order) and to fp64 within rounding."""
    from efg_amd.operators.linear import column_sum, relu_backward_column_sum

    g = torch.Generator().manual_seed(rows + cols)
    grad = torch.randn(rows, cols, generator=g).cuda()
    y = torch.relu(torch.randn(rows, cols, generator=g)).cuda()
    masked, sums = relu_backward_column_sum(grad, y)
    want = torch.ops.aten.threshold_backward(grad, y, 0)
    assert torch.equal(masked, want)
    assert torch.equal(sums, column_sum(want))
    ref = want.double().sum(0)
    assert float((sums.double() - ref).abs().max()) <= 1e-5 * float(want.abs().double().sum(0).max()) + 1e-12


@pytest.mark.parametrize("same_input", [False, True])
def test_self_attention_in_proj_matches_sliced_linears(same_input):
    """operators/linear.py:self_attention_in_proj against F.linear on slices of the packed parameters (what
    nn.MultiheadAttention computes): same values, one [3c, c] / [3c] gradient without slice-backward nodes."""
    import torch.nn.functional as F

    from efg_amd.operators.linear import self_attention_in_proj

    torch.manual_seed(4)
    c = 256
    w = (torch.randn(3 * c, c, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(3 * c, device="cuda").requires_grad_(True)
    v_in = torch.randn(2, 1240, c, device="cuda", requires_grad=True)
    qk_in = v_in if same_input else torch.randn(2, 1240, c, device="cuda", requires_grad=True)
    qk, v = self_attention_in_proj(qk_in, v_in, w, b)
    assert type(qk.grad_fn).__name__ == "SelfAttentionInProjFunctionBackward"
    wr, br, vr = (t.detach().clone().requires_grad_(True) for t in (w, b, v_in))
    qr = vr if same_input else qk_in.detach().clone().requires_grad_(True)
    qk_ref, v_ref = F.linear(qr, wr[:2 * c], br[:2 * c]), F.linear(vr, wr[2 * c:], br[2 * c:])
    assert torch.equal(qk, qk_ref) and torch.equal(v, v_ref)
    u1, u2 = torch.randn_like(qk), torch.randn_like(v)
    ((qk * u1).sum() + (v * u2).sum()).backward()
    ((qk_ref * u1).sum() + (v_ref * u2).sum()).backward()
    assert torch.allclose(w.grad, wr.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(b.grad, br.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(v_in.grad, vr.grad, rtol=1e-5, atol=1e-5)
    if not same_input:
        assert torch.allclose(qk_in.grad, qr.grad, rtol=1e-5, atol=1e-5)


def test_linear_refuses_nothing_on_cpu():
    """Host tensors run the plain PyTorch layer (dense layers are PyTorch plumbing, not a HIP op)."""
    from efg_amd.operators.linear import Linear

    m = Linear(8, 4)
    x = torch.randn(3, 8, requires_grad=True)
    m(x).sum().backward()
    assert m.bias.grad is not None


def test_conv3x3_fixed_order_weight_gradient(dev):
    """EFG_DETERMINISTIC=1: the dense 3 x 3 convolution (forward, data gradient, weight gradient) as nine fixed-order GEMMs
    each over the padded channels-last maps (operators/conv2d.py) -- equal to fp64 F.conv2d to fp32 rounding, and the same
    bits every time."""
    from efg_amd.operators.conv2d import conv3x3

    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 45, 52, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(96, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)
    b = torch.randn(96, generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(2, 96, 45, 52, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref.backward(gy.double())
    rx, rw, rb = x.grad.clone(), w.grad.clone(), b.grad.clone()
    outs = []
    for trial in range(3):
        x.grad = w.grad = b.grad = None
        y = conv3x3(x, w, b)
        y.backward(gy)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()))
    y, gx, gw, gb = outs[0]
    for got, exp, name in ((y, ref.detach(), "out"), (gx, rx, "grad x"), (gw, rw, "grad w"), (gb, rb, "grad b")):
        scale = float(exp.abs().max())
        assert float((got.double() - exp.double()).abs().max()) < 2e-5 * scale, name
    for o in outs[1:]:   # forward, data and weight gradient all run as fixed-order GEMMs: the same bits every time
        assert torch.equal(o[0], y) and torch.equal(o[1], gx), "output / data gradient differ between two identical passes"
        assert torch.equal(o[2], gw), "weight gradient differs between two identical backward passes"
