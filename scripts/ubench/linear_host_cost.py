"""Host time of one small F.linear / addmm call with TunableOp's lookup on and off (GPU box)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(2, 1240, 256, device=dev)
w = torch.randn(256, 256, device=dev)
b = torch.randn(256, device=dev)


def host_us(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


print("tunable off: F.linear %.1f us/call, addmm %.1f us/call" % (host_us(lambda: F.linear(x, w, b)),
                                                               host_us(lambda: torch.addmm(b, x.view(-1, 256), w.t()))))
engine.configure_hip_runtime()
if hasattr(engine, "_enable_tuned_gemms"):
    engine._enable_tuned_gemms()
import torch.cuda.tunable as tunable  # noqa: E402

tunable.enable(True)
tunable.tuning_enable(False)
path = os.path.join(ROOT, "efg_amd", "tuned", "gemm_gfx950.csv")
if os.path.exists(path):
    tunable.set_filename(path)
print("tunable on : F.linear %.1f us/call, addmm %.1f us/call" % (host_us(lambda: F.linear(x, w, b)),
                                                               host_us(lambda: torch.addmm(b, x.view(-1, 256), w.t()))))
tunable.enable(False)
for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        print("tunable off, preferred %-8s: F.linear %.1f us/call, addmm %.1f us/call, mm %.1f us/call" % (
            lib, host_us(lambda: F.linear(x, w, b)), host_us(lambda: torch.addmm(b, x.view(-1, 256), w.t())),
            host_us(lambda: torch.mm(x.view(-1, 256), w))))
    except Exception as exc:  # noqa: BLE001
        print("tunable", lib, "failed:", str(exc)[:100])
e = torch.empty(2480, 256, device=dev)
print("tunable-free baselines: torch.empty %.1f us, x + 1 %.1f us, relu %.1f us" % (
    host_us(lambda: torch.empty(2480, 256, device=dev)), host_us(lambda: x + 1), host_us(lambda: torch.relu(x))))
tunable.enable(True)
for lib in ("cublas", "cublaslt"):
    torch.backends.cuda.preferred_blas_library(lib)
    print("tunable ON, preferred %-8s: F.linear %.1f us/call, mm %.1f us/call, mm^T %.1f us/call" % (
        lib, host_us(lambda: F.linear(x, w, b)), host_us(lambda: torch.mm(x.view(-1, 256), w)),
        host_us(lambda: torch.mm(x.view(-1, 256).t(), x.view(-1, 256)))))
