"""Which GEMM shapes run on which library kernel inside one training step (torch.profiler with shapes) -- GPU box."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch.profiler import ProfilerActivity, profile
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(pool[0])
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm") and e.kernels:
        for k in e.kernels:
            key = (e.name, str(e.input_shapes), k.name[:60])
            agg[key][0] += 1
            agg[key][1] += k.duration
for (name, shapes, kern), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.1f us x %3d  %-11s %-70s %s" % (us / n, n, name, shapes[:70], kern))
