"""What would ONE launch for the main + shortcut strided convolutions of a residual stage buy?  (sparse_net.py:125-165: same
input, same rulebook.)  Upper bound without writing the kernel: a single SparseConv3d with the two weight sets concatenated
along Cout IS that launch for the forward (twice the n-slices over the same tiles) and for the weight gradient, and its data
gradient is the K-concatenated product.  GPU box.
    python scripts/ubench/conv_pair.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efg_amd.spconv as spconv  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402
from efg_amd.operators import voxelize_batch  # noqa: E402
from efg_amd.spconv import core  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
pts = [torch.from_numpy(make_scene(2000 + i)[0]).to(dev) for i in range(2)]
vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
x = spconv.SparseConvTensor(vox["voxel_mean"], vox["coordinates"], [41, 1504, 1504], 2)
x = spconv.SparseConv3d(5, 4, 3, 2, padding=1, bias=False).to(dev)(x)   # the stem's level


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


for cin, cout in ((32, 64), (64, 128), (128, 256)):
    feat = torch.randn(x.features.shape[0], cin, device=dev)
    xin = x.replace_feature(feat)
    one = spconv.SparseConv3d(cin, cout, 3, 2, padding=1, bias=False).to(dev)
    two = spconv.SparseConv3d(cin, 2 * cout, 3, 2, padding=1, bias=False).to(dev)
    y1, y2 = one(xin), two(xin)
    rb = one._rulebook(xin)[0]
    w1 = one.weight.reshape(cout, rb.kvol, cin).contiguous().detach()
    w2 = two.weight.reshape(2 * cout, rb.kvol, cin).contiguous().detach()
    g1, g2 = torch.randn_like(y1.features), torch.randn_like(y2.features)
    for _ in range(2):
        f1 = timeit(lambda: core._conv_forward(feat, w1, None, rb, one.weight))
        f2 = timeit(lambda: core._conv_forward(feat, w2, None, rb, two.weight))
        d1 = timeit(lambda: core._conv_dgrad(g1, w1, rb, one.weight))
        d2 = timeit(lambda: core._conv_dgrad(g2, w2, rb, two.weight))
        q1 = timeit(lambda: core._conv_wgrad(feat, g1, rb))
        q2 = timeit(lambda: core._conv_wgrad(feat, g2, rb))
        print("%3d -> %3d (+ %3d), rows %6d -> %6d: forward 2 x %.1f = %.1f us, one launch %.1f | dgrad 2 x %.1f = %.1f, one %.1f | "
              "wgrad 2 x %.1f = %.1f, one %.1f" % (cin, cout, cout, rb.m_in, rb.m_out, f1, 2 * f1, f2, d1, 2 * d1, d2, q1, 2 * q1, q2))
    x = y1.replace_feature(y1.features)   # next level's sites
