"""Host vs device time of the phases of one training step (GPU box): host wall-clock to ISSUE forward / backward /
optimizer against the device time between HIP events recorded at the same points.
    python scripts/ubench/tf_timeline.py [--prepared]            TrajectoryFormer (4 samples)
    python scripts/ubench/tf_timeline.py --model conquer|centerpoint   (2 scenes x 180k points)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer  # noqa: E402
from efg_amd.tracking.synthetic import synthetic_tracking_batch  # noqa: E402
from efg_amd.tracking.trajectoryformer import TrajectoryFormer  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else "trajectoryformer"
if which == "trajectoryformer":
    tr = Trainer(config=os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"), device=dev, seed=0,
                 model_cls=TrajectoryFormer, max_iters=10000)
    np.random.seed(1000)
    pool = [synthetic_tracking_batch(7000 + 100 * p, 4, device=dev, n_points=180000, n_objects=60, n_false=20) for p in range(4)]
else:
    from efg_amd.centerpoint.voxelnet import VoxelNet
    from efg_amd.engine import synthetic_batch

    kw = {"config": os.path.join(ROOT, "configs", "centerpoint_waymo_voxelnet.yaml"), "model_cls": VoxelNet} if which == "centerpoint" else {}
    tr = Trainer(device=dev, seed=0, max_iters=10000, **kw)
    pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(4)]
prepared = "--prepared" in sys.argv and which == "trajectoryformer"
rows = []


def step(batch, keep):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    if prepared:
        batch = tr.model.prepare([([dict(s[0])], i) for s, i in batch])
        torch.cuda.synchronize()
    h = [time.perf_counter()]
    ev[0].record()
    tr.optimizer.zero_grad(set_to_none=True)
    loss_dict = tr.wrapped(batch)
    losses = torch.stack([v for v in loss_dict.values() if torch.is_tensor(v) and v.requires_grad]).sum()
    h.append(time.perf_counter())
    ev[1].record()
    losses.backward()
    h.append(time.perf_counter())
    ev[2].record()
    params = [p for p in tr.model.parameters() if p.grad is not None]
    if tr.grad_clipper is not None:
        torch.nn.utils.clip_grad_norm_(params, **dict(tr.grad_clipper.params))
    tr.optimizer.step()
    if tr.lr_scheduler is not None:
        tr.lr_scheduler.step()
    h.append(time.perf_counter())
    ev[3].record()
    if keep:
        rows.append((h, ev))


for i in range(5):
    step(pool[i % 4], False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    step(pool[i % 4], True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
host = np.array([[(h[k + 1] - h[k]) * 1e3 for k in range(3)] for h, _ in rows]).mean(0)
gpu = np.array([[ev[k].elapsed_time(ev[k + 1]) for k in range(3)] for _, ev in rows]).mean(0)
print("%s: %.2f ms/step wall; host issue time forward %.2f backward %.2f optimizer %.2f (sum %.2f); device time between the "
      "same points forward %.2f backward %.2f optimizer %.2f (sum %.2f)"
      % (which + (": preparation outside the timed phases" if prepared else ""), wall, *host,
         host.sum(), *gpu, gpu.sum()))
