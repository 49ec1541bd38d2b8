"""Which aten ops (and input shapes) launch a given ATen kernel family inside one training step?  (GPU box)

    python scripts/ubench/op_stacks.py <kernel-name-substring> [...]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(pool[0])
    torch.cuda.synchronize()
pats = sys.argv[1:] or ["elementwise_kernel_manual_unroll"]
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    for k in (e.kernels or []):
        if any(p in k.name for p in pats):
            key = (e.name, str(e.input_shapes)[:110], "bwd" if e.thread != prof.events()[0].thread else "fwd")
            agg[key][0] += 1
            agg[key][1] += k.duration
top = int(os.environ.get("OP_STACKS_TOP", "40"))
for (name, shapes, th), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%7.1f us total %3d x  %s %-28s %s" % (us, n, th, name, shapes))
