"""Does the BLAS behind torch.matmul compute fp32 GEMMs in fp32?  Error vs fp64 of the layouts a Linear layer uses
(forward NT, dgrad NN, wgrad TN) through rocBLAS and hipBLASLt.  GPU box."""
import os
import sys

import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)


def err(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    for (m, n, k) in ((1024, 256, 256), (256, 256, 1024), (70688, 256, 256), (256, 256, 70688), (256, 1024, 70688)):
        a = torch.randn(m, k, device=dev)
        b = torch.randn(n, k, device=dev)
        nt = err(a @ b.t(), a.double() @ b.double().t())                       # forward: x W^T
        nn_ = err(a @ b.t().contiguous(), a.double() @ b.double().t())         # NN
        at = torch.randn(k, m, device=dev)
        bt = torch.randn(k, n, device=dev)
        tn = err(at.t() @ bt, at.double().t() @ bt.double())                   # wgrad: G^T X
        lin = torch.nn.functional.linear(a, b)
        print("%-9s m=%6d n=%5d k=%6d   NT %.1e  NN %.1e  TN %.1e  F.linear %.1e" % (lib, m, n, k, nt, nn_, tn,
                                                                                 err(lin, a.double() @ b.double().t())))
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "fp32 precision", torch.get_float32_matmul_precision())
print({k: v for k, v in os.environ.items() if "BLAS" in k or "TF32" in k or "TUNABLE" in k})
