"""Do the dgrad and the wgrad of one sparse-conv layer overlap usefully on two streams?  (both are ~one occupancy
wave at 2 scenes per GPU, each with the matrix pipe ~45 % busy.)  GPU box.
    python scripts/ubench/conv_overlap.py --level res3"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efg_amd.spconv as spconv  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402
from efg_amd.operators import voxelize_batch  # noqa: E402
from efg_amd.spconv import core  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", default="res3")
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
pts = [torch.from_numpy(make_scene(2000 + i)[0]).to(dev) for i in range(2)]
vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
x = spconv.SparseConvTensor(vox["voxel_mean"], vox["coordinates"], [41, 1504, 1504], 2)
chan = {"stem": 32, "res2": 64, "res3": 128, "res4": 256}
cin = 5
for name in ["stem", "res2", "res3", "res4"]:
    x = spconv.SparseConv3d(cin if name == "stem" else 4, 4, 3, 2, padding=1, bias=False).to(dev)(x)
    cin = 4
    if name == args.level:
        break
c = chan[args.level]
conv = spconv.SubMConv3d(c, c, 3, padding=1, bias=False, indice_key="k").to(dev)
feat = torch.randn(x.features.shape[0], c, device=dev)
xin = x.replace_feature(feat)
y = conv(xin)
rb = conv._rulebook(xin)[0]
go = torch.randn_like(y.features)
w = conv.weight.reshape(c, rb.kvol, c).contiguous().detach()
side = torch.cuda.Stream()


def seq():
    core._conv_dgrad(go, w, rb)
    core._conv_wgrad(feat, go, rb)


def par():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        core._conv_wgrad(feat, go, rb)
    core._conv_dgrad(go, w, rb)
    main.wait_stream(side)


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / args.iters


for _ in range(2):
    print("%s rows %d: dgrad alone %.1f us, wgrad alone %.1f us, sequential %.1f us, two streams %.1f us" % (
        args.level, rb.m_out, timeit(lambda: core._conv_dgrad(go, w, rb)), timeit(lambda: core._conv_wgrad(feat, go, rb)),
        timeit(seq), timeit(par)))
