"""`efg.operators.ms_deform_attn.MSDeformAttn` (the nn.Module named in the north star) against the REFERENCE module
imported in place (tests/golden/msdeform_module.npz from scripts/make_golden_msdeform.py: the reference
`MSDeformAttn`, n_levels 4 / n_points 4, evaluated with the reference's `ms_deform_attn_core_pytorch`): the reference
state dict loads by name, a fresh module has the reference's initial sampling offsets, and output / input gradients /
parameter gradients agree for 2-column and 4-column reference points with a padding mask.
CPU: oracle sampling op (pins the module's host logic); GPU: the HIP kernels."""
import contextlib

import numpy as np
import pytest
import torch

from conftest import golden


def _module(device):
    from efg_amd.operators.ms_deform_attn import MSDeformAttn

    g = golden("msdeform_module.npz")
    m = MSDeformAttn(d_model=64, n_levels=4, n_heads=8, n_points=4)
    # a fresh module starts from the reference's star-shaped offsets (ms_deform_attn.py:107-121)
    np.testing.assert_allclose(m.sampling_offsets.bias.detach().numpy(), g["init_offsets_bias"], rtol=0, atol=1e-6)
    assert float(m.sampling_offsets.weight.abs().max()) == 0.0 and float(m.attention_weights.weight.abs().max()) == 0.0
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")}, strict=True)
    return m.to(device), g


def _check(device, ctx, fwd_tol, bwd_tol):
    m, g = _module(device)
    t = lambda k: torch.from_numpy(g[k]).to(device)  # noqa: E731
    for tag in ("pt", "box"):
        q = t(tag + "_query").requires_grad_(True)
        x = t(tag + "_input").requires_grad_(True)
        m.zero_grad()
        with ctx():
            out = m(q, t(tag + "_ref"), x, t("shapes"), t("start"), t(tag + "_mask"))
            out.backward(t(tag + "_go"))
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag + "_out"], rtol=0, atol=fwd_tol)
        for name, got in (("_gq", q.grad), ("_gx", x.grad), ("_g_offsets_w", m.sampling_offsets.weight.grad),
                          ("_g_attn_w", m.attention_weights.weight.grad), ("_g_value_w", m.value_proj.weight.grad)):
            want = g[tag + name]
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=bwd_tol * np.abs(want).max(),
                                       err_msg=tag + name)
    with pytest.raises(ValueError), ctx():
        m(q, t("pt_ref")[..., :1], x, t("shapes"), t("start"))


def test_msdeform_module_cpu(oracle_mod):
    from oracle import cpu_backend

    _check(torch.device("cpu"), cpu_backend.install, 1e-5, 1e-4)


@pytest.mark.gpu
def test_msdeform_module_gpu(dev):
    _check(dev, contextlib.nullcontext, 1e-5, 1e-4)
