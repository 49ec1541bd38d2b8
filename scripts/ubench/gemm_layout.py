"""Data-gradient products of the encoder-sized Linear layers: dX = dY . W with W [out, in] row-major is an "NN" product for the
GEMM library; with a transposed copy Wt [in, out] it is dX = dY . Wt^T, the "NT" layout of the forward pass.  Which one is faster
at [70688 x K] x [K x N], fp32, with the tuned solutions (efg_amd/tuned/gemm_gfx950.csv)?  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import use_tuned_gemms  # noqa: E402

use_tuned_gemms()
dev = torch.device("cuda:0")
M = 70688


def timeit(fn, iters=30):
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


for out_f, in_f in ((256, 256), (232, 256), (1024, 256), (256, 1024), (256, 384)):
    w = torch.randn(out_f, in_f, device=dev)          # nn.Linear weight
    wt = w.t().contiguous()
    dy = torch.randn(M, out_f, device=dev)
    acc = torch.randn(M, in_f, device=dev)
    nn_ = timeit(lambda: dy.mm(w))
    nt_ = timeit(lambda: dy.mm(wt.t()))
    nn_acc = timeit(lambda: acc.addmm_(dy, w))
    nt_acc = timeit(lambda: acc.addmm_(dy, wt.t()))
    tr = timeit(lambda: w.t().contiguous())
    gf = 2.0 * M * out_f * in_f / 1e9
    print("dX[%d,%d] = dY[%d,%d] . W[%d,%d]: NN %.1f us (%.0f TF/s), NT on a transposed copy %.1f us (%.0f TF/s); accumulating (addmm_) NN %.1f / NT %.1f; "
          "transpose itself %.1f us" % (M, in_f, M, out_f, out_f, in_f, nn_, gf / nn_ * 1e3, nt_, gf / nt_ * 1e3, nn_acc, nt_acc, tr))
