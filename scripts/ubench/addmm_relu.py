"""FFN first layer [70688, 256] x [256, 1024] + bias + ReLU: addmm then relu vs torch._addmm_activation (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import use_tuned_gemms  # noqa: E402

use_tuned_gemms()
dev = torch.device("cuda:0")
x = torch.randn(70688, 256, device=dev)
w = torch.randn(1024, 256, device=dev) * 0.05
b = torch.randn(1024, device=dev)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = torch.relu(torch.addmm(b, x, w.t()))
fused = torch._addmm_activation(b, x, w.t(), use_gelu=False)
print("max diff", float((ref - fused).abs().max()), "equal", bool(torch.equal(ref, fused)))
print("addmm            %.1f us" % timeit(lambda: torch.addmm(b, x, w.t())))
print("addmm + relu     %.1f us" % timeit(lambda: torch.relu(torch.addmm(b, x, w.t()))))
print("addmm + relu_    %.1f us" % timeit(lambda: torch.relu_(torch.addmm(b, x, w.t()))))
print("_addmm_activation %.1f us" % timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)))
