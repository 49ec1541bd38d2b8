"""cProfile of TrajectoryFormer._prepare (the parameter-free half of the step) on the GPU box."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer  # noqa: E402
from efg_amd.tracking.synthetic import synthetic_tracking_batch  # noqa: E402
from efg_amd.tracking.trajectoryformer import TrajectoryFormer  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(config=os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"), device=dev, seed=0,
             model_cls=TrajectoryFormer, max_iters=10000)
np.random.seed(1000)
pool = [synthetic_tracking_batch(7000 + 100 * p, 4, device=dev, n_points=180000, n_objects=60, n_false=20) for p in range(4)]
m = tr.model
m.load_pretrain_motionencoder()
m.batch_size = 4


def f(n):
    for i in range(n):
        with torch.no_grad():
            m._prepare(pool[i % 4])
    torch.cuda.synchronize()


f(4)
cProfile.run("f(20)", "/tmp/tfprep.prof")
st = pstats.Stats("/tmp/tfprep.prof")
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
