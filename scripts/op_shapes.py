"""Which aten ops (by input shape) own the non-GEMM device time of a training step?  (GPU box)

    python scripts/op_shapes.py [n_rows]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for s in range(STEPS):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = [e for e in rows if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
tot = sum(e.self_device_time_total for e in rows)
print("self device time, all ops: %.2f ms/step" % (tot / STEPS / 1e3))
for e in rows[:n]:
    print("%7.3f ms %4d  %-42s %s" % (e.self_device_time_total / STEPS / 1e3, e.count // STEPS, e.key[:42],
                                     str(e.input_shapes)[:150]))
