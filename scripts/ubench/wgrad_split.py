"""Weight-gradient GEMM of the encoder's 256-wide Linear layers: [256, K] x [K, 256], K = 70 688 rows.
Library split-K (one mm) against an explicit batched split (bmm over K chunks + a sum).  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import use_tuned_gemms  # noqa: E402

use_tuned_gemms()
dev = torch.device("cuda:0")


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


for K, cin, cout in [(70688, 256, 256), (70688, 256, 1024), (70688, 1024, 256), (70688, 256, 200), (70688, 256, 32), (70688, 384, 256), (2480, 256, 256), (2480, 256, 1024)]:
    x = torch.randn(K, cin, device=dev)
    g = torch.randn(K, cout, device=dev)
    ref = x.t().mm(g)
    print("K=%d cin=%d cout=%d   mm: %.1f us" % (K, cin, cout, timeit(lambda: x.t().mm(g))))
    for s in (8, 16, 32, 47):
        xs, gs = x.view(s, K // s, cin), g.view(s, K // s, cout)
        f = lambda: torch.bmm(xs.transpose(1, 2), gs).sum(0)
        err = float((f() - ref).abs().max() / ref.abs().max())
        f2 = lambda: torch.bmm(gs.transpose(1, 2), xs).sum(0)  # [cout, cin]: the parameter's own layout
        print("    bmm split %2d + sum: %.1f us   (rel diff %.1e)     g^T x form: %.1f us" % (s, timeit(f), err, timeit(f2)))
