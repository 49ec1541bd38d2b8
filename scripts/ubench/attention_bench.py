"""csrc/attention.hip against PyTorch's fp32 SDPA (aotriton) at the point encoder's shape [1232, 128 tokens, 4 heads x 64]:
forward and backward time per call and the fraction of the fp32 MFMA roof (GPU box)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.operators.attention import _SelfAttention  # noqa: E402

dev = torch.device("cuda:0")
B, S, H = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1232, 128, 4)))
qkv = torch.randn(B, S, 3 * H * 64, device=dev, requires_grad=True)
w = torch.randn(B, S, H * 64, device=dev)


def sdpa(x):
    q, k, v = (t.reshape(B, S, H, 64).transpose(1, 2) for t in x.chunk(3, dim=-1))
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, H * 64)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


flops_fwd = 4.0 * B * H * S * S * 64
for name, fn in (("attention.hip", lambda x: _SelfAttention.apply(x, H)), ("SDPA fp32", sdpa)):
    fwd = timed(lambda: fn(qkv.detach()))
    y = fn(qkv)
    both = timed(lambda: torch.autograd.grad(fn(qkv), qkv, w))
    bwd = both - fwd
    print("%-14s forward %7.1f us (%.1f TFLOP/s, %.2f of 157.3)   backward %7.1f us (%.1f TFLOP/s algorithmic 2.5x forward)"
          % (name, fwd, flops_fwd / fwd / 1e6, flops_fwd / fwd / 1e6 / 157.3, bwd, 2.5 * flops_fwd / bwd / 1e6))
