"""One sparse-conv layer of the res18 backbone at its real geometry (2 synthetic 180k-point scenes), timed alone:
    python scripts/ubench/conv_layer.py [--level stem|res2|res3|res4] [--kind subm|down|out] [--pass fwd|dgrad|wgrad|all]
                                        [--iters N]
Under `rocprofv3 --pmc ...` the counters of the conv kernel then belong to exactly one layer.  GPU box."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efg_amd.spconv as spconv  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402
from efg_amd.operators import voxelize_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", default="res2")
ap.add_argument("--kind", default="subm")
ap.add_argument("--pass", dest="which", default="fwd")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--scenes", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
pts = [torch.from_numpy(make_scene(2000 + i)[0]).to(dev) for i in range(args.scenes)]
vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
x = spconv.SparseConvTensor(vox["voxel_mean"], vox["coordinates"], [41, 1504, 1504], args.scenes)
chan = {"stem": 32, "res2": 64, "res3": 128, "res4": 256}
order = ["stem", "res2", "res3", "res4"]
cin = 5
for name in order:  # geometry only: strided convs with tiny channel counts down to the requested level
    down = spconv.SparseConv3d(cin if name == "stem" else 4, 4, 3, 2, padding=1, bias=False).to(dev)
    if name == args.level and args.kind == "down":
        break
    x = down(x)
    cin = 4
    if name == args.level:
        break
c = chan[args.level]
if args.kind == "subm":
    conv = spconv.SubMConv3d(c, c, 3, padding=1, bias=False, indice_key="k").to(dev)
    feat = torch.randn(x.features.shape[0], c, device=dev)
elif args.kind == "down":
    prev = {"stem": 5, "res2": 32, "res3": 64, "res4": 128}[args.level]
    conv = spconv.SparseConv3d(prev, c, 3, 2, padding=1, bias=False).to(dev)
    feat = torch.randn(x.features.shape[0], prev, device=dev)
else:
    conv = spconv.SparseConv3d(c, c, (3, 1, 1), (2, 1, 1), padding=(1, 0, 0), bias=False).to(dev)
    feat = torch.randn(x.features.shape[0], c, device=dev)
xin = x.replace_feature(feat.requires_grad_(True))
y = conv(xin)
rb = conv._rulebook(xin)[0]
pairs = rb.num_pairs()
go = torch.randn_like(y.features)
flops = 2.0 * pairs * conv.in_channels * conv.out_channels


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / args.iters


w = conv.weight.reshape(conv.out_channels, rb.kvol, conv.in_channels).contiguous()
from efg_amd.spconv import core  # noqa: E402

res = {}
if args.which in ("fwd", "all"):
    res["fwd"] = timeit(lambda: core._conv_forward(feat.detach(), w.detach(), None, rb))
if args.which in ("dgrad", "all"):
    res["dgrad"] = timeit(lambda: core._conv_dgrad(go, w.detach(), rb))
if args.which in ("wgrad", "all"):
    res["wgrad"] = timeit(lambda: core._conv_wgrad(feat.detach(), go, rb))
print("%s %s m_in=%d m_out=%d cin=%d cout=%d pairs/row %.2f  %s" % (
    args.level, args.kind, rb.m_in, rb.m_out, conv.in_channels, conv.out_channels, pairs / max(rb.m_out, 1),
    "  ".join("%s %.1f us (%.1f TF/s)" % (k, v, flops / v / 1e6) for k, v in res.items())))
