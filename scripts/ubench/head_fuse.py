"""CenterPoint's SepHead ($CP1/center_head.py:18-52): five stacks conv3x3(64 -> 64) + BN + ReLU + conv3x3(64 -> k) on the same
[2, 64, 188, 188] map, k = 2 / 1 / 3 / 2 / 3.  What would running them as ONE conv3x3(64 -> 320) + BN(320) + ReLU + ONE
conv3x3(320 -> 11) with a block-diagonal weight buy, forward + backward?  GPU box.
    python scripts/ubench/head_fuse.py"""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.operators.batchnorm import run_sequential  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
KS = (2, 1, 3, 2, 3)


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def stack(cin, mid, k):
    return nn.Sequential(nn.Conv2d(cin, mid, 3, padding=1), nn.BatchNorm2d(mid, eps=1e-3, momentum=0.01), nn.ReLU(),
                         nn.Conv2d(mid, k, 3, padding=1)).to(dev).to(memory_format=torch.channels_last)


x = torch.randn(2, 64, 188, 188, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
heads = [stack(64, 64, k) for k in KS]
fused = stack(64, 64 * len(KS), sum(KS))
mask = torch.zeros_like(fused[3].weight)
o = 0
for i, k in enumerate(KS):
    mask[o:o + k, 64 * i:64 * (i + 1)] = 1
    o += k
with torch.no_grad():
    fused[3].weight.mul_(mask)
gos = [torch.randn(2, k, 188, 188, device=dev).contiguous(memory_format=torch.channels_last) for k in KS]
go = torch.cat(gos, 1).contiguous(memory_format=torch.channels_last)


def sep_fwd():
    return [run_sequential(h, x) for h in heads]


def sep_both():
    ys = sep_fwd()
    torch.autograd.backward(ys, gos)


def fused_fwd():
    return run_sequential(fused, x)


def fused_both():
    y = fused_fwd()
    y.backward(go)
    fused[3].weight.grad.mul_(mask)


for _ in range(2):
    a, b = timeit(sep_fwd), timeit(sep_both)
    c, d = timeit(fused_fwd), timeit(fused_both)
    print("five stacks: forward %.1f us, forward + backward %.1f us | one 320-wide stack: forward %.1f us, forward + backward %.1f us"
          % (a, b, c, d), flush=True)
