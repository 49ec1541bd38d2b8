"""The loader-side chain ON the device in front of every training step (row n3): ground-truth paste from the HBM-resident
database, flip, rotation, scaling, range filter, shuffle -> ConQueR step.  Reports ms per step with / without the chain,
the chain alone, and the same chain in NumPy on one host core (what a DataLoader worker of the reference spends per
sample, without its disk reads).  GPU box."""
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.data.gpu_pipeline import DevicePoints, build_train_pipeline, run  # noqa: E402
from efg_amd.data.gt_database import DeviceGTDatabase  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, make_scene  # noqa: E402
from efg_amd.data.synthetic_db import make_database  # noqa: E402
from efg_amd.engine import Trainer  # noqa: E402

dev = torch.device("cuda:0")
np.random.seed(0)
infos, clouds = make_database(seed=7, per_class=400)
db = DeviceGTDatabase(infos, clouds, [{"VEHICLE": 15}, {"PEDESTRIAN": 10}, {"CYCLIST": 10}], min_points=5, device=dev)
chain = build_train_pipeline(PC_RANGE, database=db)
names = np.array(["VEHICLE", "PEDESTRIAN", "CYCLIST"])
raw = []
for s in range(4):
    pts, boxes, labels = make_scene(9100 + s, n_points=180000, n_boxes=12)
    raw.append((torch.from_numpy(pts).to(dev), {"gt_boxes": boxes[:, [0, 1, 2, 3, 4, 5, 8]].copy(), "gt_names": names[labels - 1],
                                                "difficulty": np.zeros(len(labels), np.int64),
                                                "num_points_in_gt": np.full(len(labels), 50, np.int64)}))


def sample(i):
    pts, ann = raw[i % len(raw)]
    cloud, info = run(chain, DevicePoints(pts.clone()), {"annotations": copy.deepcopy(ann)})
    a = info["annotations"]
    a["labels"] = np.array([list(names).index(n) + 1 for n in a["gt_names"]], np.int64)
    a["gt_boxes"] = a["gt_boxes"].astype(np.float32)
    return ({"points": cloud}, {"annotations": a})


tr = Trainer(device=dev, seed=0)
fixed = [[sample(2 * p), sample(2 * p + 1)] for p in range(2)]
for w in range(5):
    tr.step(fixed[w % 2])


def timed(fn, n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


step_only = timed(lambda i: tr.step(fixed[i % 2]))
both = timed(lambda i: tr.step([sample(2 * i), sample(2 * i + 1)]))
chain_only = timed(lambda i: (sample(2 * i), sample(2 * i + 1)))
from efg_amd.data.loader import DeviceLoader  # noqa: E402

with DeviceLoader(sample, batch_size=2, length=45, device=dev) as loader:
    for _ in range(5):
        tr.step(next(loader))
    ahead = timed(lambda i: tr.step(next(loader)), n=40)
print("2 scenes x 180k points: step on prepared batches %.2f ms; loader chain inline on the main stream + step %.2f ms; the "
      "chain alone (2 samples) %.2f ms; chain one batch ahead on the DeviceLoader thread/stream + step %.2f ms"
      % (step_only, both, chain_only, ahead))
print("points per sample after paste + filter: %d; ground-truth boxes %d" % (fixed[0][0][0]["points"].shape[0],
                                                                           len(fixed[0][0][1]["annotations"]["gt_boxes"])))
