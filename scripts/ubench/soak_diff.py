"""Which kernels get slower as ConQueR trains?  Kernel time per step (torch profiler, 4 steps) early (after 60 steps) and
late (after 560 steps) on the same 4-batch pool, largest differences first.  GPU box."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0, max_iters=700)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(4)]


def kernel_times():
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(4):
            tr.step(pool[i % 4])
        torch.cuda.synchronize()
    acc = collections.defaultdict(float)
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            acc[e.name[:90]] += e.device_time / 4e3   # ms per step
    return acc


step = 0
for target, tag in ((60, "early"), (560, "late")):
    while step < target:
        tr.step(pool[step % 4])
        step += 1
    if tag == "early":
        early = kernel_times()
    else:
        late = kernel_times()
    step += 4
print("kernel ms/step: early %.2f, late %.2f" % (sum(early.values()), sum(late.values())))
rows = sorted(set(early) | set(late), key=lambda k: -(late.get(k, 0.0) - early.get(k, 0.0)))
for k in rows[:14]:
    print("%+7.3f ms  (%.3f -> %.3f)  %s" % (late.get(k, 0) - early.get(k, 0), early.get(k, 0), late.get(k, 0), k))
