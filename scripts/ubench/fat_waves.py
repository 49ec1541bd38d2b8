"""Per-wave clock stamps of conv_fat_kernel on one res18 layer (efg_spconv_fat_debug): how long do the waves of a launch
live, and what does a wave's time depend on?   python scripts/ubench/fat_waves.py [--level res2] [--kind subm]   (GPU box)"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efg_amd.spconv as spconv  # noqa: E402
from efg_amd import _lib as L  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402
from efg_amd.operators import voxelize_batch  # noqa: E402
from efg_amd.spconv import core  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", default="res2")
args = ap.parse_args()
dev = torch.device("cuda:0")
pts = [torch.from_numpy(make_scene(2000 + i)[0]).to(dev) for i in range(2)]
vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
x = spconv.SparseConvTensor(vox["voxel_mean"], vox["coordinates"], [41, 1504, 1504], 2)
chan = {"res2": 64, "res3": 128, "res4": 256}
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
rb = conv._rulebook(xin)[0]
w = conv.weight.reshape(c, rb.kvol, c).contiguous().detach()
for _ in range(3):
    core._conv_forward(feat, w, None, rb)
buf = torch.zeros(2048 * 16, dtype=torch.int64, device=dev)
L.check(L.lib().efg_spconv_fat_debug(buf.data_ptr()))
core._conv_forward(feat, w, None, rb)
torch.cuda.synchronize()
L.check(L.lib().efg_spconv_fat_debug(None))
d = buf.cpu().numpy().reshape(2048, 16)
d = d[d[:, 1] > 0]
t0 = np.zeros(len(d), dtype=np.int64)
for xcd in range(16):   # the counters of the XCCs are not synchronised: times relative to the XCC's first wave
    m = d[:, 7] == xcd
    if m.any():
        t0[m] = d[m, 0].min()
start, end = d[:, 0] - t0, d[:, 1] - t0
life = end - start
tick = 1.0 / 2400.0  # s_memtime counts shader cycles (~2.4 GHz); every XCC has its own counter
print("%s: %d waves; launch spans %.1f us; wave start %.1f..%.1f us, end %.1f..%.1f us (median %.1f), life mean %.1f / max %.1f us" % (
    args.level, len(d), end.max() * tick, start.min() * tick, start.max() * tick, end.min() * tick, end.max() * tick,
    np.median(end) * tick, life.mean() * tick, life.max() * tick))
work = d[:, 2] * 2 + d[:, 3]
A = np.stack([np.ones(len(d)), d[:, 2], d[:, 3], d[:, 4], d[:, 5]], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, life * tick, rcond=None)
print("life ~ %.1f us + %.2f us per full step + %.2f per half step + %.2f per unit + %.2f per cut unit" % tuple(coef))
print("steps per wave: full %.1f half %.1f units %.1f cut %.1f; 16-row-tile steps min %d max %d" % (
    d[:, 2].mean(), d[:, 3].mean(), d[:, 4].mean(), d[:, 5].mean(), work.min(), work.max()))
names = ["prologue", "partial store + drain", "ticket", "sum of partials", "row stores", "open unit (incl. first)"]
print("per wave, mean us: " + ", ".join("%s %.2f" % (n, d[:, 8 + i].mean() * tick) for i, n in enumerate(names)))
for xcd in range(0):
    m = d[:, 7] == xcd
    if m.any():
        print("  XCC %d: %4d waves, end median %.1f max %.1f us, life mean %.1f" % (xcd, m.sum(), np.median(end[m]) * tick, end[m].max() * tick, life[m].mean() * tick))
