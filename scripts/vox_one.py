"""One hard-voxelization call shape, repeated (GPU box): the workload of scripts/vox_traffic.sh and of ad-hoc timing.

    python scripts/vox_one.py <points> <sweeps> <scenes> <calls>      prints "calls voxels alg_bytes us_per_call" """
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402
from efg_amd.operators.voxelize import _hard_voxelize_launch  # noqa: E402

n, sw, nb, calls = (int(a) for a in sys.argv[1:5])
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
run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(calls):
    run()
e1.record()
torch.cuda.synchronize()
m = int(num.sum())
alg = 4 * f * pts.shape[0] + m * (4 * 5 * f + 16 + 4 + 4 * f)
print("VOX calls %d voxels %d alg_bytes %d us_per_call %.1f" % (calls + 1, m, alg, e0.elapsed_time(e1) * 1e3 / calls))
