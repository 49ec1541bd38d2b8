"""Latency of the inference paths (ConQueR / Voxel-DETR `forward` in eval mode, CenterPoint
`VoxelNet` in eval mode, TrajectoryFormer's online tracker `forward_inference`), random-init weights, synthetic scenes of
the BASELINE size.  GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.centerpoint.voxelnet import VoxelNet  # noqa: E402
from efg_amd.config import load_config  # noqa: E402
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402
from efg_amd.tracking import TrajectoryFormer  # noqa: E402
from efg_amd.tracking.synthetic import make_tracking_sequence  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n, warm=3):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for name, kw in (("ConQueR (1000 queries)", {}),
                 ("Voxel-DETR", {"config": os.path.join(ROOT, "configs", "voxeldetr_waymo_res18.yaml")}),
                 ("CenterPoint VoxelNet", {"config": os.path.join(ROOT, "configs", "centerpoint_waymo_voxelnet.yaml"),
                                           "model_cls": VoxelNet})):
    tr = Trainer(device=dev, seed=0, **kw)
    tr.model.eval()
    pool = [synthetic_batch(5000 + 10 * p, 1, n_points=180000, device=dev) for p in range(4)]
    with torch.no_grad():
        ms = timed(lambda i: tr.model(pool[i % 4]), 20)
        out = tr.model(pool[0])
    n_det = len(out[0]["scores"]) if isinstance(out, (list, tuple)) and "scores" in out[0] else -1
    print("%-24s eval forward, 1 scene x 180k points: %6.2f ms (%d boxes out)" % (name, ms, n_det))
    tr.close()
    del tr

cfg = load_config(os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"),
                  {"model.device": str(dev), "task": "val", "model.eval_class": "VEHICLE"})
torch.manual_seed(0)
model = TrajectoryFormer(cfg).eval()
seq = make_tracking_sequence(seed=3, frames=40, n_objects=40, n_ground=150000, per_object=300)
with torch.no_grad():
    for item in seq[:5]:
        model([item])
    torch.cuda.synchronize()
    t = time.perf_counter()
    tracks = 0
    for item in seq[5:]:
        tracks = len(model([item])[0]["track_ids"])
    torch.cuda.synchronize()
print("TrajectoryFormer online tracker, %d objects, %d-point sweeps: %6.2f ms per frame (%d tracks in the last frame)"
      % (40, seq[0][0][0]["points"].shape[0], (time.perf_counter() - t) / (len(seq) - 5) * 1e3, tracks))
