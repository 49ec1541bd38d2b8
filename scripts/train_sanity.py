"""Does the engine train?  N steps on a small fixed pool of synthetic batches: the total loss must fall and every
parameter stay finite.  GPU box.
    python scripts/train_sanity.py [--model conquer|voxeldetr|centerpoint|trajectoryformer] [--steps 120]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="conquer")
ap.add_argument("--steps", type=int, default=120)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = lambda n: os.path.join(ROOT, "configs", n)  # noqa: E731
if args.model == "centerpoint":
    from efg_amd.centerpoint import VoxelNet

    tr = Trainer(config=cfg("centerpoint_waymo_voxelnet.yaml"), device=dev, seed=0, model_cls=VoxelNet, max_iters=args.steps)
    pool = [synthetic_batch(4000 + 10 * p, 2, device=dev) for p in range(3)]
elif args.model == "trajectoryformer":
    from efg_amd.tracking import TrajectoryFormer
    from efg_amd.tracking.synthetic import synthetic_tracking_batch

    np.random.seed(0)
    tr = Trainer(config=cfg("trajectoryformer_waymo_centerpoint.yaml"), device=dev, seed=0, model_cls=TrajectoryFormer,
                 max_iters=args.steps)
    pool = [synthetic_tracking_batch(5000 + 10 * p, 2, device=dev, n_points=60000, n_objects=30, n_false=8) for p in range(3)]
else:
    tr = Trainer(config=None if args.model == "conquer" else cfg("voxeldetr_waymo_res18.yaml"), device=dev, seed=0,
                 max_iters=args.steps)
    pool = [synthetic_batch(2000 + 10 * p, 2, device=dev) for p in range(3)]
hist = []
for s in range(args.steps):
    _, total = tr.step(pool[s % len(pool)])
    hist.append(total.detach())
vals = torch.stack(hist).float().cpu().numpy()
ok = all(torch.isfinite(p).all().item() for p in tr.model.parameters())
first, last = vals[:6].mean(), vals[-6:].mean()
print("%s: %d steps on %d fixed batches (OneCycle over %d steps): total loss %.3f -> %.3f (first / last 6-step mean), "
      "min %.3f, parameters finite: %s" % (args.model, args.steps, len(pool), args.steps, first, last, vals.min(), ok))
print("every 10th step:", " ".join("%.2f" % v for v in vals[::10]))
assert ok and last < first, "the loss did not fall"
