"""csrc/attention.hip against fp64 attention (and PyTorch's fp32 SDPA as the yardstick): forward, log-sum-exp and the
gradient of the fused in-projection output, for full and ragged sequence lengths."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(qkv, heads, dtype):
    x = qkv.detach().to(dtype).requires_grad_(True)
    b, s, c3 = x.shape
    q, k, v = (t.reshape(b, s, heads, 64).transpose(1, 2) for t in x.chunk(3, dim=-1))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(64), -1)
    lse = torch.logsumexp(q @ k.transpose(-1, -2) / math.sqrt(64), -1)
    out = (att @ v).transpose(1, 2).reshape(b, s, c3 // 3)
    return x, out, lse


@pytest.mark.parametrize("batch,seq,heads", [(37, 128, 4), (5, 100, 4), (3, 72, 2), (4, 16, 1), (2, 1, 4), (3, 33, 8)])
def test_forward_backward_match_fp64(batch, seq, heads):
    from efg_amd import _lib as L
    from efg_amd.operators.attention import _SelfAttention, fused

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(batch * 1000 + seq)
    qkv = (torch.randn(batch, seq, 3 * heads * 64, generator=g) * 1.5).to(dev).requires_grad_(True)
    w = torch.randn(batch, seq, heads * 64, generator=g).to(dev)
    assert fused(qkv, heads)
    out = _SelfAttention.apply(qkv, heads)
    (out * w).sum().backward()
    x64, out64, lse64 = _reference(qkv, heads, torch.float64)
    (out64 * w.double()).sum().backward()
    x32, out32, _ = _reference(qkv, heads, torch.float32)
    (out32 * w).sum().backward()

    def err(a, ref):
        return float((a.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))

    # as accurate as PyTorch's own fp32 evaluation of the same formula (within 4x), and tight in absolute terms
    assert err(out, out64) <= max(4 * err(out32, out64), 2e-6), (err(out, out64), err(out32, out64))
    assert err(qkv.grad, x64.grad) <= max(4 * err(x32.grad, x64.grad), 5e-6), (err(qkv.grad, x64.grad), err(x32.grad, x64.grad))
    # the saved log-sum-exp, through the C ABI
    c = heads * 64
    x = qkv.detach()
    lse = torch.empty(batch, heads, seq, device=dev)
    o2 = torch.empty(batch, seq, c, device=dev)
    L.check(L.lib().efg_attention_fwd_f32(x.data_ptr(), seq * 3 * c, 3 * c, x.data_ptr() + 4 * c, x.data_ptr() + 8 * c,
                                          seq * 3 * c, 3 * c, batch, seq, seq, heads, 1 / math.sqrt(64), L.ptr(o2), L.ptr(lse),
                                          L.stream()))
    assert torch.equal(o2, out.detach())
    assert err(lse, lse64) <= 1e-6


@pytest.mark.parametrize("batch,sq,sk,heads", [(41, 1, 128, 4), (7, 5, 77, 2), (3, 128, 16, 4), (2, 40, 128, 1)])
def test_cross_attention_matches_fp64(batch, sq, sk, heads):
    from efg_amd.operators.attention import cross_attention_kv, fused_cross

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(batch * 100 + sq + sk)
    c = heads * 64
    q = torch.randn(batch, sq, c, generator=g).to(dev).requires_grad_(True)
    kv = (torch.randn(batch, sk, 2 * c, generator=g) * 1.3).to(dev).requires_grad_(True)
    w = torch.randn(batch, sq, c, generator=g).to(dev)
    assert fused_cross(q, kv, heads)
    out = cross_attention_kv(q, kv, heads)
    (out * w).sum().backward()

    def ref(dtype):
        q_, kv_ = q.detach().to(dtype).requires_grad_(True), kv.detach().to(dtype).requires_grad_(True)
        k_, v_ = kv_.chunk(2, -1)

        def split(t):
            return t.reshape(batch, t.shape[1], heads, 64).transpose(1, 2)

        att = torch.softmax(split(q_) @ split(k_).transpose(-1, -2) / 8.0, -1)
        o = (att @ split(v_)).transpose(1, 2).reshape(batch, sq, c)
        (o * w.to(dtype)).sum().backward()
        return o.detach(), q_.grad, kv_.grad

    o64, dq64, dkv64 = ref(torch.float64)
    o32, dq32, dkv32 = ref(torch.float32)

    def err(a, r):
        return float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-30))

    assert err(out, o64) <= max(4 * err(o32, o64), 2e-6)
    assert err(q.grad, dq64) <= max(4 * err(dq32, dq64), 5e-6)
    assert err(kv.grad, dkv64) <= max(4 * err(dkv32, dkv64), 5e-6)


def test_matches_sdpa_in_the_point_encoder_layer():
    """`attend` with the kernel against `attend` through SDPA, gradients to the parameters included."""
    import os

    from torch import nn

    from efg_amd.tracking.layers import attend

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mha = nn.MultiheadAttention(256, 4).to(dev)
    x = torch.randn(61, 128, 256, device=dev)
    token = torch.randn(61, 1, 256, device=dev)
    for cross in (False, True):
        res = {}
        for mode in ("1", "0"):
            os.environ["EFG_ATTENTION"] = mode
            try:
                mha.zero_grad()
                xi = x.clone().requires_grad_(True)
                ti = token.clone().requires_grad_(True)
                y = attend(mha, ti, xi) if cross else attend(mha, xi, xi)
                y.square().sum().backward()
                res[mode] = (y.detach(), xi.grad, mha.in_proj_weight.grad.clone(), mha.out_proj.weight.grad.clone()) + \
                    ((ti.grad,) if cross else ())
            finally:
                os.environ.pop("EFG_ATTENTION", None)
        for a, b in zip(res["1"], res["0"]):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


def test_bad_arguments_are_reported():
    from efg_amd import _lib as L

    dev = torch.device("cuda:0")
    qkv = torch.zeros(1, 129, 192, device=dev)
    out = torch.zeros(1, 129, 64, device=dev)
    lse = torch.zeros(1, 1, 129, device=dev)
    with pytest.raises(RuntimeError, match="seq"):
        L.check(L.lib().efg_attention_fwd_f32(L.ptr(qkv), 129 * 192, 192, qkv.data_ptr() + 256, qkv.data_ptr() + 512, 129 * 192,
                                              192, 1, 129, 129, 1, 0.125, L.ptr(out), L.ptr(lse), L.stream()))
    with pytest.raises(RuntimeError, match="strides"):
        L.check(L.lib().efg_attention_fwd_f32(L.ptr(qkv), 64 * 190, 190, qkv.data_ptr() + 256, qkv.data_ptr() + 512, 64 * 190,
                                              190, 1, 64, 64, 1, 0.125, L.ptr(out), L.ptr(lse), L.stream()))


def _long_reference(qk, v, mask, heads, dtype):
    qk_, v_ = qk.detach().to(dtype).requires_grad_(True), v.detach().to(dtype).requires_grad_(True)
    b, s, c = v_.shape
    q, k = qk_.chunk(2, -1)

    def split(t):
        return t.reshape(b, s, heads, 32).transpose(1, 2)

    logits = split(q) @ split(k).transpose(-1, -2) / math.sqrt(32)
    if mask is not None:
        logits = logits.masked_fill(mask, float("-inf"))
    out = (torch.softmax(logits, -1) @ split(v_)).transpose(1, 2).reshape(b, s, c)
    return qk_, v_, out, torch.logsumexp(logits, -1)


@pytest.mark.parametrize("batch,seq,heads,masked", [(2, 1240, 8, True), (2, 1000, 8, False), (3, 129, 2, True), (1, 64, 1, True),
                                                    (2, 5, 4, False), (1, 333, 3, True)])
def test_long_attention_matches_fp64(batch, seq, heads, masked):
    from efg_amd.operators.attention import long_self_attention, pack_mask

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seq * 10 + heads)
    c = heads * 32
    qk = (torch.randn(batch, seq, 2 * c, generator=g) * 1.4).to(dev).requires_grad_(True)
    v = torch.randn(batch, seq, c, generator=g).to(dev).requires_grad_(True)
    w = torch.randn(batch, seq, c, generator=g).to(dev)
    mask = None
    if masked:   # denoising-style: blocks that only see themselves + random holes, every query keeps itself
        grp = torch.randint(0, 4, (seq,), generator=g)
        mask = (grp[:, None] != grp[None, :]) | (torch.rand(seq, seq, generator=g) < 0.1)
        mask[torch.arange(seq), torch.arange(seq)] = False
        mask = mask.to(dev)
    bits = pack_mask(mask)
    out = long_self_attention(qk, v, bits, heads)
    (out * w).sum().backward()
    qk64, v64, out64, lse64 = _long_reference(qk, v, mask, heads, torch.float64)
    (out64 * w.double()).sum().backward()
    qk32, v32, out32, _ = _long_reference(qk, v, mask, heads, torch.float32)
    (out32 * w).sum().backward()

    def err(a, ref):
        return float((a.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))

    assert err(out, out64) <= max(4 * err(out32, out64), 2e-6), (err(out, out64), err(out32, out64))
    assert err(qk.grad, qk64.grad) <= max(4 * err(qk32.grad, qk64.grad), 5e-6), (err(qk.grad, qk64.grad), err(qk32.grad, qk64.grad))
    assert err(v.grad, v64.grad) <= max(4 * err(v32.grad, v64.grad), 5e-6)


def test_pack_mask_bits():
    from efg_amd.operators.attention import pack_mask

    g = torch.Generator().manual_seed(3)
    m = torch.rand(70, 70, generator=g) < 0.5
    bits = pack_mask(m.cuda()).cpu()
    assert bits.shape == (70, 3) and bits.dtype == torch.int32
    for q in (0, 13, 69):
        for k in (0, 31, 32, 63, 64, 69):
            assert bool((int(bits[q, k // 32]) >> (k % 32)) & 1) == bool(m[q, k])


def test_decoder_layer_self_attention_matches_the_module():
    """TransformerDecoderLayer._self_attention: kernel path vs the nn.MultiheadAttention call it replaces."""
    import os

    from efg_amd.detection3d.transformer import TransformerDecoderLayer
    from efg_amd.operators.attention import pack_mask

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layer = TransformerDecoderLayer(256, 8, 1, 1024, 0.0).to(dev)
    x, pos = torch.randn(2, 1100, 256, device=dev), torch.randn(2, 1100, 256, device=dev)
    grp = torch.randint(0, 3, (1100,), device=dev)
    mask = grp[:, None] != grp[None, :]
    res = {}
    for mode in ("1", "0"):
        os.environ["EFG_ATTENTION"] = mode
        try:
            layer.zero_grad()
            xi, pi = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
            y = layer._self_attention(xi + pi, xi, mask, pack_mask(mask) if mode == "1" else None)
            y.square().sum().backward()
            res[mode] = (y.detach(), xi.grad, pi.grad, layer.self_attn.in_proj_weight.grad.clone(),
                         layer.self_attn.in_proj_bias.grad.clone(), layer.self_attn.out_proj.weight.grad.clone())
        finally:
            os.environ.pop("EFG_ATTENTION", None)
    for a, b in zip(res["1"], res["0"]):
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max())
