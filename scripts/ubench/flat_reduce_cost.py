"""Host and device cost of FlatGradientAllReduce.reduce() per step (1 rank, RCCL backend).  GPU box."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29519")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
tr = Trainer(device="cuda:0", seed=0, ddp=True)
pool = [synthetic_batch(2000 + 100 * p, 2, device="cuda:0") for p in range(2)]
gs = tr.grad_sync
orig = gs.reduce
host, devt = [], []


def timed():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    e0.record()
    orig()
    e1.record()
    host.append(time.perf_counter() - t)
    devt.append((e0, e1))


gs.reduce = timed
for s in range(26):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
print("reduce(): host %.2f ms, device span %.2f ms per step; %d tensors, %.1f MB" % (
    1e3 * sum(host[6:]) / len(host[6:]), sum(a.elapsed_time(b) for a, b in devt[6:]) / len(devt[6:]),
    len(gs.params), gs.flat.numel() * 4 / 1e6))
dist.destroy_process_group()
