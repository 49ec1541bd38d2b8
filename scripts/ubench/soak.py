"""Soak: N training steps (ConQueR; --model centerpoint | trajectoryformer) on a 4-batch pool; step time of the first / last 50 steps and the allocator's
peak after 100 steps vs at the end (a leak or a slow drift shows up in either).  GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
dev = torch.device("cuda:0")
which = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else "conquer"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if which == "trajectoryformer":
    import numpy as np

    from efg_amd.tracking import TrajectoryFormer
    from efg_amd.tracking.synthetic import synthetic_tracking_batch

    tr = Trainer(config=os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"), device=dev, seed=0,
                 model_cls=TrajectoryFormer, max_iters=n + 10)
    np.random.seed(0)
    pool = [synthetic_tracking_batch(7000 + 100 * p, 4, device=dev, n_points=180000, n_objects=60, n_false=20) for p in range(4)]
elif which == "centerpoint":
    from efg_amd.centerpoint import VoxelNet

    tr = Trainer(config=os.path.join(ROOT, "configs", "centerpoint_waymo_voxelnet.yaml"), device=dev, seed=0, model_cls=VoxelNet,
                 max_iters=n + 10)
    pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(4)]
else:
    tr = Trainer(device=dev, seed=0, max_iters=n + 10)
    pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(4)]
frozen = "--frozen" in sys.argv   # same work, weights never change: separates clock / thermal drift from training dynamics
if frozen:
    for g in tr.optimizer.param_groups:
        g["lr"] = 0.0
        g["weight_decay"] = 0.0
    tr.lr_scheduler = None
for i in range(10):
    tr.step(pool[i % 4])
torch.cuda.synchronize()
times, peak100 = [], None
for i in range(n):
    t = time.perf_counter()
    loss = tr.step(pool[i % 4])[1]
    if i % 50 == 49:
        torch.cuda.synchronize()
    times.append(time.perf_counter() - t)
    if i == 99:
        peak100 = torch.cuda.max_memory_allocated()
torch.cuda.synchronize()
first, last = sum(times[50:100]) / 50 * 1e3, sum(times[-50:]) / 50 * 1e3
print("ms/step per 50 steps (mean / max):", " ".join("%.1f/%.0f" % (sum(times[k:k + 50]) / 50 * 1e3, max(times[k:k + 50]) * 1e3)
                                                      for k in range(0, n, 50)))
print(which + ": %d steps: %.2f ms/step over steps 50-100, %.2f over the last 50; peak allocated %.2f GB after 100 steps, %.2f GB at the "
      "end; last loss %.3f, finite %s" % (n, first, last, peak100 / 2 ** 30, torch.cuda.max_memory_allocated() / 2 ** 30,
                                          float(loss), bool(torch.isfinite(loss))))
