"""Wall time of the first steps of a fresh process (what a small --warmup would leave inside the timed region)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

t0 = time.perf_counter()
tr = Trainer(device="cuda:0", overrides={"model.transformer.num_queries": 1000}, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device="cuda:0") for p in range(2)]
torch.cuda.synchronize()
print("build + data: %.1f s" % (time.perf_counter() - t0))
for s in range(10):
    t = time.perf_counter()
    tr.step(pool[s % 2])
    torch.cuda.synchronize()
    print("step %d: %.1f ms" % (s, 1000 * (time.perf_counter() - t)), flush=True)
