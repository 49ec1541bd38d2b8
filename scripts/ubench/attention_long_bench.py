"""csrc/attention.hip long-sequence kernels against nn.MultiheadAttention's SDPA path at ConQueR's decoder shape
[2 scenes, 1240 queries, 8 heads x 32] with a denoising-style boolean mask (GPU box)."""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.detection3d.transformer import TransformerDecoderLayer  # noqa: E402
from efg_amd.operators.attention import pack_mask  # noqa: E402

dev = torch.device("cuda:0")
B, S = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (2, 1240)))
torch.manual_seed(0)
layer = TransformerDecoderLayer(256, 8, 1, 1024, 0.0).to(dev)
x = torch.randn(B, S, 256, device=dev, requires_grad=True)
pos = torch.randn(B, S, 256, device=dev)
grp = torch.cat([torch.zeros(1000 if S > 1000 else S // 2, dtype=torch.long), 1 + torch.arange(S - (1000 if S > 1000 else S // 2)) // 60]).to(dev)
mask = grp[:, None] != grp[None, :]
bits = pack_mask(mask)
w = torch.randn(B, S, 256, device=dev)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


for mode, name in (("1", "attention.hip (q|k + v projections, kernel, out projection)"), ("0", "nn.MultiheadAttention (SDPA fp32)")):
    os.environ["EFG_ATTENTION"] = mode
    b = bits if mode == "1" else None
    with torch.no_grad():
        fwd = timed(lambda: layer._self_attention(x + pos, x, mask, b))
    both = timed(lambda: torch.autograd.grad(layer._self_attention(x + pos, x, mask, b), x, w))
    print("%-62s forward %6.1f us   forward + backward (to the input) %6.1f us" % (name, fwd, both))
