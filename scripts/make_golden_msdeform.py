"""Golden vectors for the `MSDeformAttn` nn.Module: the reference module (efg/operators/ms_deform_attn.py:85-198)
imported in place on CPU, with its `MSDeformAttnFunction` routed to the reference's own pure-PyTorch core
(`ms_deform_attn_core_pytorch`, :55-76) because `efg._C` has no CPU kernel.  Default geometry of the module
(n_levels 4, n_points 4), both reference-point formats.  Saves weights (reference state-dict names), inputs, the
freshly initialised sampling-offset bias, outputs and gradients to tests/golden/msdeform_module.npz."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REF)
c_stub = types.ModuleType("efg._C")
c_stub.__getattr__ = lambda n: (_ for _ in ()).throw(AttributeError(n)) if n.startswith("__") else (lambda *a, **k: None)
import efg  # noqa: E402

sys.modules["efg._C"] = c_stub
efg._C = c_stub
import efg.operators.ms_deform_attn as ref  # noqa: E402


class _Core:
    @staticmethod
    def apply(value, shapes, start, loc, attn, step):
        return ref.ms_deform_attn_core_pytorch(value, shapes.tolist(), loc, attn)


ref.MSDeformAttnFunction = _Core
torch.manual_seed(0)
D, L, H, P = 64, 4, 8, 4
m = ref.MSDeformAttn(d_model=D, n_levels=L, n_heads=H, n_points=P)
save = {"init_offsets_bias": m.sampling_offsets.bias.detach().clone()}
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    m.sampling_offsets.weight.copy_(torch.randn(m.sampling_offsets.weight.shape, generator=g) * 0.05)
    m.attention_weights.weight.copy_(torch.randn(m.attention_weights.weight.shape, generator=g) * 0.1)
    m.attention_weights.bias.copy_(torch.randn(m.attention_weights.bias.shape, generator=g) * 0.1)
for k, v in m.state_dict().items():
    save["w::" + k] = v.detach().clone()
shapes = torch.tensor([[8, 9], [4, 5], [2, 3], [1, 1]])
start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
S, N, LQ = int(shapes.prod(1).sum()), 2, 23
for tag, width in (("pt", 2), ("box", 4)):
    q = torch.randn(N, LQ, D, generator=g).requires_grad_(True)
    x = torch.randn(N, S, D, generator=g).requires_grad_(True)
    refp = torch.rand(N, LQ, L, width, generator=g)
    if width == 4:
        refp[..., 2:] = refp[..., 2:] * 0.4 + 0.05
    mask = torch.zeros(N, S, dtype=torch.bool)
    mask[1, -7:] = True
    m.zero_grad()
    out = m(q, refp, x, shapes, start, mask)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save.update({tag + "_query": q.detach(), tag + "_input": x.detach(), tag + "_ref": refp, tag + "_mask": mask,
                 tag + "_out": out.detach(), tag + "_go": go, tag + "_gq": q.grad.clone(), tag + "_gx": x.grad.clone(),
                 tag + "_g_offsets_w": m.sampling_offsets.weight.grad.clone(),
                 tag + "_g_attn_w": m.attention_weights.weight.grad.clone(),
                 tag + "_g_value_w": m.value_proj.weight.grad.clone()})
save["shapes"], save["start"] = shapes, start
out_path = os.path.join(ROOT, "tests", "golden", "msdeform_module.npz")
np.savez_compressed(out_path, **{k: v.numpy() for k, v in save.items()})
print("saved", out_path, os.path.getsize(out_path) // 1024, "KiB")
