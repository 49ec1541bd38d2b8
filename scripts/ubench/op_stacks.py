import os, sys, torch
sys.path.insert(0, "/root/repo")
from torch.profiler import ProfilerActivity, profile
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(pool[0])
    torch.cuda.synchronize()
want = {"aten::copy_": [[2, 256, 188, 188], [2, 188, 188, 256], [3, 80, 1000, 256]], "aten::add": [[2, 35344, 256]], "aten::add_": [[2, 35344, 256]], "aten::fill_": [[2, 35344, 8, 32]]}
for e in prof.events():
    if e.name in want and e.input_shapes and e.input_shapes[0] in want[e.name]:
        st = [s for s in (e.stack or []) if "efg_amd" in s or "autograd" in s.lower()][:4]
        kd = sum(k.duration for k in e.kernels) if e.kernels else 0
        print(e.name, e.input_shapes[:2], "%.0f us" % kd, "thread", e.thread, "|", " <- ".join(s.split("/")[-1] for s in st))
