"""Is the momentum decoder replayed from a HIP graph, and what does that do to host / device time?  GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

tr = Trainer(device="cuda:0", seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device="cuda:0") for p in range(2)]
t = tr.model.transformer
orig = t._run_gt_decoder
host, dev = [], []


def timed(*a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    out = orig(*a)
    e1.record()
    host.append(time.perf_counter() - t0)
    dev.append((e0, e1))
    return out


t._run_gt_decoder = timed
for s in range(16):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
print("graphs:", len(t.__dict__.get("_gt_graphs", {})), "disabled:", getattr(t, "_gt_graph_off", False))
print("gt decoder per step: host %.2f ms, device span %.2f ms" % (
    1e3 * sum(host[6:]) / len(host[6:]), sum(a.elapsed_time(b) for a, b in dev[6:]) / len(dev[6:])))
