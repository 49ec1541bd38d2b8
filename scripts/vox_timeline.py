"""Phase timeline of the binned voxelizer's four kernels (GPU box): per kernel and marker, when the first / median / last
workgroup passed it, in microseconds from the first marker of the call.  csrc/voxelize_bins.hip `mark()`.

    python scripts/vox_timeline.py [n_points] [sweeps] [scenes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efg_amd import _lib  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402
from efg_amd.operators.voxelize import _hard_voxelize_launch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 180000
sw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
scenes = [torch.from_numpy(make_scene(2000 + i, n_points=n, n_sweeps=sw)[0]).to(dev) for i in range(nb)]
pts = torch.cat(scenes)
offs = [0]
for s in scenes:
    offs.append(offs[-1] + s.shape[0])
f, mv = pts.shape[1], 120000 if sw == 1 else 200000
cap = nb * mv
voxels = torch.empty((cap, 5, f), device=dev)
coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
npv = torch.empty(cap, dtype=torch.int32, device=dev)
mean = torch.empty((cap, f), device=dev)
num = torch.zeros(nb, dtype=torch.int32, device=dev)
run = lambda: _hard_voxelize_launch(pts, offs, VOXEL_SIZE, PC_RANGE, 5, mv, voxels, coors, npv, num, mean)  # noqa: E731
for _ in range(5):
    run()
torch.cuda.synchronize()
words = _lib.lib().efg_hard_voxelize_debug_timeline(None)
buf = torch.zeros(words, dtype=torch.int64, device=dev)
_lib.lib().efg_hard_voxelize_debug_timeline(_lib.ptr(buf))
NAMES = {0: "bin_count", 1: "bin_scan", 2: "bin_scatter", 3: "first", 5: "rank", 4: "write"}
MARK = {(0, 0): "start", (0, 1): "tile staged", (0, 2): "LDS aggregation", (0, 3): "group atomics + places",
        (1, 0): "start", (1, 1): "chunk published", (1, 2): "look-back", (1, 3): "end",
        (2, 0): "start", (2, 1): "end", (3, 0): "start", (3, 1): "bin work", (5, 0): "start", (5, 1): "end", (4, 0): "start", (4, 1): "big: sorted", (4, 2): "big: written", (4, 3): "small: sorted",
        (4, 4): "small: written"}
for rep in range(3):
    buf.zero_()
    run()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(6, 8, -1)
    t0 = t[t > 0].min()
    print("--- call %d: %d scenes x %d points, %d voxels" % (rep, nb, n, int(num.sum())))
    for k in (0, 1, 2, 3, 5, 4):
        for m in range(8):
            v = t[k, m]
            v = v[v > 0]
            if v.size:
                us = (v - t0) / 100.0
                print("  %-12s %-24s wgs %6d  first %8.1f  median %8.1f  last %8.1f us" % (NAMES[k], MARK.get((k, m), m), v.size, us.min(),
                                                                                         np.median(us), us.max()))
_lib.lib().efg_hard_voxelize_debug_timeline(None)
