"""GPU-busy time of a training step: sum of the device durations of all kernels / memsets / copies, per step
(torch.profiler).  Less noisy than wall time for A/B-ing a change (GPU box).

    python scripts/gpu_busy.py [steps]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(6):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for s in range(STEPS):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
dev_events = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
busy = sum(e.time_range.end - e.time_range.start for e in dev_events)
print("gpu busy: %.3f ms/step in %d launches/step" % (busy / STEPS / 1e3, len(dev_events) // STEPS))
