"""Host time of one GEMM call (tiny shapes, launch-bound) through the library paths PyTorch can take.  GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
x = torch.randn(320, 256, device=dev)
w = torch.randn(256, 256, device=dev)
b = torch.randn(256, device=dev)


def host_us(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / n * 1e6


def report(tag):
    print("%-34s linear %5.1f us  addmm %5.1f us  mm %5.1f us  relu %5.1f us" % (
        tag, host_us(lambda: torch.nn.functional.linear(x, w, b)), host_us(lambda: torch.addmm(b, x, w.t())),
        host_us(lambda: x.mm(w)), host_us(lambda: torch.relu(x))), flush=True)


print("preferred blas:", torch.backends.cuda.preferred_blas_library())
report("default, TunableOp off")
import torch.cuda.tunable as tunable  # noqa: E402

tunable.enable(True)
tunable.tuning_enable(False)
report("TunableOp on (no entries)")
from efg_amd.engine import use_tuned_gemms  # noqa: E402

use_tuned_gemms()
report("TunableOp on (committed csv)")
tunable.enable(False)
for lib in ("cublas", "cublaslt"):
    torch.backends.cuda.preferred_blas_library(lib)
    report("preferred_blas_library=%s" % lib)
