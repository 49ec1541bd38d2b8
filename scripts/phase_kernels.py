"""Kernel launches and device time per phase of one training step (GPU box)."""
import os, sys
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
    tr.step(pool[0])
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
ranges = [e for e in evs if e.name.startswith("efg::")]
ops = [e for e in evs if e.kernels]
print("%-26s %8s %10s %10s" % ("range", "kernels", "dev_ms", "cpu_ms"))
for r in sorted(ranges, key=lambda e: e.time_range.start):
    inside = [o for o in ops if o.time_range.start >= r.time_range.start and o.time_range.end <= r.time_range.end and o.thread == r.thread]
    nk = sum(len(o.kernels) for o in inside)
    dt = sum(k.duration for o in inside for k in o.kernels) / 1e3
    print("%-26s %8d %10.2f %10.2f" % (r.name, nk, dt, (r.time_range.end - r.time_range.start) / 1e3))
# backward runs on the autograd thread: everything not on the main thread
main_thread = ranges[0].thread
bw = [o for o in ops if o.thread != main_thread]
print("%-26s %8d %10.2f" % ("(autograd thread)", sum(len(o.kernels) for o in bw), sum(k.duration for o in bw for k in o.kernels) / 1e3))
from collections import Counter
c = Counter()
t = Counter()
for o in bw:
    c[o.name] += len(o.kernels); t[o.name] += sum(k.duration for k in o.kernels) / 1e3
print("top backward ops by kernel count:")
for n, k in c.most_common(25):
    print("   %-60s %6d kernels %8.2f ms" % (n[:60], k, t[n]))
fw = [o for o in ops if o.thread == main_thread]
c = Counter(); t = Counter()
for o in fw:
    c[o.name] += len(o.kernels); t[o.name] += sum(k.duration for k in o.kernels) / 1e3
print("top forward-thread ops by kernel count:")
for n, k in c.most_common(25):
    print("   %-60s %6d kernels %8.2f ms" % (n[:60], k, t[n]))
