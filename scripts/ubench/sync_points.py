"""Where does a training step synchronise the host with the device?  torch.cuda.set_sync_debug_mode("warn") around one step
(after warm-up) and the distinct Python locations of the warnings.  GPU box.
    python scripts/ubench/sync_points.py [conquer|centerpoint|voxeldetr|trajectoryformer]"""
import os
import sys
import traceback
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "conquer"
dev = torch.device("cuda:0")
cfg = lambda n: os.path.join(ROOT, "configs", n)  # noqa: E731
if which == "trajectoryformer":
    from efg_amd.tracking import TrajectoryFormer
    from efg_amd.tracking.synthetic import synthetic_tracking_batch

    tr = Trainer(config=cfg("trajectoryformer_waymo_centerpoint.yaml"), device=dev, seed=0, model_cls=TrajectoryFormer, max_iters=1000)
    np.random.seed(0)
    pool = [synthetic_tracking_batch(7000 + 100 * p, 4, device=dev, n_points=180000, n_objects=60, n_false=20) for p in range(3)]
    prep = lambda b: tr.model.prepare([([dict(s[0])], i) for s, i in b])  # noqa: E731  (the loader's collate, outside the watch)
else:
    kw = {}
    if which == "centerpoint":
        from efg_amd.centerpoint import VoxelNet

        kw = {"config": cfg("centerpoint_waymo_voxelnet.yaml"), "model_cls": VoxelNet}
    elif which == "voxeldetr":
        kw = {"config": cfg("voxeldetr_waymo_res18.yaml")}
    tr = Trainer(device=dev, seed=0, max_iters=1000, **kw)
    pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(3)]
    prep = lambda b: b  # noqa: E731
for i in range(4):
    tr.step(prep(pool[i % 3]))
torch.cuda.synchronize()
batch = prep(pool[1])
torch.cuda.synchronize()
seen = {}
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    stack = [f for f in traceback.extract_stack() if "/efg_amd/" in f.filename]
    where = " <- ".join("%s:%d" % (os.path.relpath(f.filename, ROOT), f.lineno) for f in reversed(stack[-3:]))
    if not where:   # no frame of ours: the innermost frames of whatever called
        where = "(outside efg_amd) " + " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno)
                                                   for f in reversed(traceback.extract_stack()[-6:-1]))
    seen[where] = seen.get(where, 0) + 1


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
tr.step(batch)
torch.cuda.set_sync_debug_mode("default")
warnings.showwarning = orig
print("%s: %d synchronising calls in one step" % (which, sum(seen.values())))
for where, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d x %s" % (n, where))
