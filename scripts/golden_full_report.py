"""Print the error of every quantity of the full-model golden (tests/test_model_full_golden.py) -- CPU (oracle ops)
or GPU (HIP ops): activations, losses, gradients, as max-abs-error / max-abs-value."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_full_golden as T  # noqa: E402

dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
# A/B toggles (GPU): --toggle sdp_math | unfused_box | unfused_loss | (env switches EFG_FUSED_* are read by the ops)
toggles = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--toggle"]
if "sdp_math" in toggles:
    torch.backends.cuda.enable_flash_sdp(False)
    torch.backends.cuda.enable_mem_efficient_sdp(False)
if "unfused_box" in toggles:
    import efg_amd.operators.box_attention_func as baf
    baf.FUSED_ENABLED = False
if "torch_box" in toggles:
    # sampling by F.grid_sample + autograd (the formulation of the reference's ms_deform_attn_core_pytorch,
    # efg/operators/ms_deform_attn.py:55-76) instead of the HIP kernels
    import torch.nn.functional as F
    import efg_amd.operators.box_attention_func as baf
    import efg_amd.detection3d.box_attention as ba
    baf.FUSED_ENABLED = False

    class _TorchBox:
        @staticmethod
        def apply(value, shapes, start, loc, attn, step):
            b, s, h, d = value.shape
            _, lq, _, L, _ = attn.reshape(attn.shape[0], attn.shape[1], attn.shape[2], attn.shape[3], -1).shape
            attn = attn.reshape(b, lq, h, L, -1)
            p = attn.shape[-1]
            loc = loc.reshape(b, lq, h, L, p, 2)
            sizes = [(int(a), int(c)) for a, c in shapes.tolist()]
            vals = value.split([a * c for a, c in sizes], dim=1)
            grids = 2 * loc - 1
            out = []
            for lid, (hh, ww) in enumerate(sizes):
                v = vals[lid].flatten(2).transpose(1, 2).reshape(b * h, d, hh, ww)
                gl = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
                out.append(F.grid_sample(v, gl, mode="bilinear", padding_mode="zeros", align_corners=False))
            a = attn.transpose(1, 2).reshape(b * h, 1, lq, L * p)
            o = (torch.stack(out, dim=-2).flatten(-2) * a).sum(-1).view(b, h * d, lq)
            return o.transpose(1, 2).contiguous()

    for mod in (baf, ba):
        if hasattr(mod, "BoxAttnFunction"):
            mod.BoxAttnFunction = _TorchBox
full = "--pruned" not in sys.argv
model, g = T._build(dev, full_graph=full)
if dev.type == "cpu":
    from oracle import cpu_backend
    ctx = cpu_backend.install()
else:
    import contextlib
    ctx = contextlib.nullcontext()
with ctx:
    cap, losses, total = T._run(model, dev)
c = lambda x: x.detach().float().cpu().numpy()  # noqa: E731
for k in ("bu_res3", "bu_res4", "fpn_p3", "src", "memory"):
    w = g[k]
    print("%-60s err/max %.2e" % (k, np.abs(c(cap[k]) - w).max() / np.abs(w).max()))
for k, v in sorted(g.items()):
    if k.startswith("loss::"):
        print("%-60s rel %.2e" % (k, abs(float(losses[k[6:]]) - float(v)) / max(abs(float(v)), 1e-9)))
params = dict(model.named_parameters())
for k, v in sorted(g.items()):
    if k.startswith("grad::"):
        got = c(params[k[6:]].grad)
        if got.size > 65536:
            got = got[:8]
        print("%-60s err/max %.2e" % (k, np.abs(got - v).max() / np.abs(v).max()))
