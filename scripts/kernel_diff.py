"""Per-kernel device time of one training step (torch.profiler), as JSON: run twice with different env and diff."""
import json, os, sys
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
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for s in range(4):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
out = {}
for e in prof.key_averages():
    if e.device_time_total > 0:
        out[e.key[:110]] = [e.count / 4, e.device_time_total / 4e3]
json.dump(out, open(sys.argv[1], "w"))
print("total ms/step", sum(v[1] for v in out.values()))
