"""Phase-level breakdown of one training step with torch.profiler (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import ProfilerActivity, profile
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for s in range(2):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = [e for e in ev if e.key.startswith("efg::")]
print("%-28s %8s %12s %12s" % ("range", "calls", "cpu_ms/step", "cuda_ms/step"))
for e in sorted(rows, key=lambda e: -e.device_time_total):
    print("%-28s %8d %12.2f %12.2f" % (e.key, e.count, e.cpu_time_total / 2e3, e.device_time_total / 2e3))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
