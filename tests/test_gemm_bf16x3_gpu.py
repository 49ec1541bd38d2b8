"""GPU: the split-precision (bf16 x 3) GEMM of the bench's A/B arm against an fp64 product -- and against the fp32
product's own error, which is the scale that matters for the parity gate (tests/test_full_size_parity_gpu.py)."""
import pytest
import torch
from conftest import force_proposals

pytestmark = pytest.mark.gpu


def _rel(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("m,k,n", [(128, 32, 128), (70688, 256, 256), (4097, 256, 1024), (3001, 1024, 256), (2049, 256, 200),
                                   (777, 200, 256), (1500, 256, 32), (640, 32, 256), (5, 64, 7)])
def test_matches_fp64(m, k, n):
    from efg_amd.operators import gemm_bf16x3 as G

    g = torch.Generator().manual_seed(m + k + n)
    a = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()     # nn.Linear weight [out, in]
    b = torch.randn(n, generator=g).cuda()
    ref = a.double() @ w.double().t() + b.double()
    out = G.gemm(a, G.pack_linear(w, transposed=False), n, bias=b)
    e3, e32 = _rel(out, ref), _rel(torch.addmm(b, a, w.t()), ref)
    # three bf16 products keep 16 significand bits per operand: a few 1e-6 of the largest entry, fp32 ~1e-7
    assert e3 < 2e-5, (e3, e32)
    # asymmetric operands: a transposed / row-col swapped result would be off by O(1), not 1e-5
    out_r = G.gemm(a, G.pack_linear(w, transposed=False), n, bias=b, relu=True)
    assert torch.equal(out_r, out.clamp_min(0))


def test_data_gradient_product_and_strided_rows():
    from efg_amd.operators import gemm_bf16x3 as G

    g = torch.Generator().manual_seed(5)
    dy = torch.randn(1000, 200, generator=g).cuda()
    w = torch.randn(200, 256, generator=g).cuda()              # [out, in]: dx = dy W
    ref = dy.double() @ w.double()
    assert _rel(G.gemm(dy, G.pack_linear(w, transposed=True), 256), ref) < 2e-5
    wide = torch.randn(1000, 512, generator=g).cuda()
    view = wide[:, 128:384]                                     # row stride 512, 16-byte aligned start
    w2 = torch.randn(64, 256, generator=g).cuda()
    assert _rel(G.gemm(view, G.pack_linear(w2, transposed=False), 64), view.double() @ w2.double().t()) < 2e-5


@pytest.mark.parametrize("m,n,k", [(70688, 256, 256), (9000, 1024, 256), (9000, 256, 1024), (4097, 200, 256), (4097, 32, 256),
                                   (31, 256, 256), (33, 8, 4)])
def test_weight_gradient_matches_fp64(m, n, k):
    from efg_amd.operators import gemm_bf16x3 as G

    gen = torch.Generator().manual_seed(m + n + k)
    g = torch.randn(m, n, generator=gen).cuda()
    x = torch.randn(m, k, generator=gen).cuda()
    ref = g.double().t() @ x.double()
    out = G.wgrad(g, x)
    assert out.shape == (n, k)
    e3, e32 = _rel(out, ref), _rel(g.t() @ x, ref)
    assert e3 < 2e-5, (e3, e32)
    assert torch.equal(out, G.wgrad(g, x))      # fixed summation order: run-to-run identical


def test_arm_passes_the_parity_gate_on_a_full_size_conquer_step(monkeypatch):
    """The gate the arm has to pass before its number may be quoted: one full-size ConQueR training step
    with the split-precision products against the same step in exact fp32 -- encoder logits within 1e-4, every loss term
    within 1e-4 relative, gradient norm within 5e-4 relative -- and the arm must actually have run."""
    import numpy as np

    from efg_amd.engine import Trainer, synthetic_batch
    from efg_amd.operators import gemm_bf16x3 as G
    from efg_amd.operators import linear as lin

    dev = torch.device("cuda:0")
    calls, wcalls = [], []
    real, real_w = G.gemm, G.wgrad
    monkeypatch.setattr(G, "gemm", lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1])
    monkeypatch.setattr(G, "wgrad", lambda *a, **k: (wcalls.append(a[0].shape), real_w(*a, **k))[1])

    def step(arm, forced):
        monkeypatch.setattr(lin, "_ARM_BF16X3", arm)
        np.random.seed(3)
        tr = Trainer(device=dev, overrides={"model.transformer.num_queries": 900}, seed=0, ddp=False)
        tr.model.noise_generator = torch.Generator().manual_seed(4321)
        seen = {}
        force_proposals(tr.model.transformer, forced)
        tr.model.transformer.register_forward_hook(lambda mod, inp, out: seen.update(
            topk=mod.enc_outputs["topk_indexes"].detach().cpu()[..., 0], logits=mod.enc_outputs["pred_logits"].detach().cpu()))
        losses, _ = tr.step(synthetic_batch(1000, 2, n_points=180000, device=dev))
        out = {k: float(v.detach()) for k, v in losses.items()}
        norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum().cpu() for p in tr.model.parameters() if p.grad is not None)))
        tr.close()
        return out, norm, seen

    ref, ref_norm, ref_seen = step(False, None)
    assert not calls
    arm, arm_norm, arm_seen = step(True, ref_seen["topk"])
    assert len(calls) >= 30 and all(s[0] >= 16384 for s in calls), len(calls)   # forward + data gradient of the long layers
    assert len(wcalls) >= 15, len(wcalls)                                        # and their weight gradients
    assert float((arm_seen["logits"] - ref_seen["logits"]).abs().max()) < 1e-4
    for k in ref:
        assert arm[k] == pytest.approx(ref[k], rel=1e-4, abs=1e-6), k
    assert arm_norm == pytest.approx(ref_norm, rel=5e-4)


def test_arm_conv3x3_matches_the_fp32_convolution():
    """operators/conv2d.py:Conv3x3ArmFunction -- the dense 3 x 3 convolution of the neck as three split-precision products per
    pass over overlapping-row views of the padded channels-last map -- against F.conv2d in fp64: forward, data gradient,
    weight and bias gradient at the accuracy of the split (a few 1e-6 of the tensor's scale)."""
    import torch.nn.functional as F

    from efg_amd.operators.conv2d import conv3x3_arm

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 37, 29, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)
    b = torch.randn(128, generator=g).to(dev).requires_grad_(True)
    y = conv3x3_arm(x, w, b)
    up = torch.randn(y.shape, generator=g).to(dev)
    (y * up).sum().backward()
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, padding=1)
    (yr * up.double()).sum().backward()

    def rel(a, ref):
        return float((a.double() - ref).abs().max() / ref.abs().max())

    assert rel(y, yr) < 2e-5 and rel(x.grad, xr.grad) < 2e-5 and rel(w.grad, wr.grad) < 2e-5 and rel(b.grad, br.grad) < 2e-5


def test_split_weights_follow_the_optimizer(monkeypatch):
    """The arm caches the split weight per parameter version: after an (in-place, fused) optimizer step the next product must
    use the NEW weight."""
    from efg_amd.operators import linear as lin

    monkeypatch.setattr(lin, "_ARM_BF16X3", True)
    g = torch.Generator().manual_seed(11)
    layer = lin.Linear(256, 256).cuda()
    opt = torch.optim.AdamW(layer.parameters(), lr=0.05, fused=True)
    x = torch.randn(20000, 256, generator=g).cuda()
    for _ in range(3):
        y = layer(x)
        ref = x.double() @ layer.weight.detach().double().t() + layer.bias.detach().double()
        assert _rel(y.detach(), ref) < 2e-5
        y.square().mean().backward()
        gw = (2.0 / y.numel()) * (y.detach().double().t() @ x.double())
        assert _rel(layer.weight.grad, gw) < 5e-5
        opt.step()
        opt.zero_grad(set_to_none=True)
