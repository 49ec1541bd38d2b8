"""Where do the sporadic slow steps come from?  Host-side phase timestamps of a pipelined run (no per-step sync)."""
import sys, time, torch
sys.path.insert(0, '.')
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device('cuda:0')
import os
DDP = os.environ.get('JITTER_DDP', '0') == '1'
if DDP:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
    torch.cuda.set_device(0); dist.init_process_group('nccl', rank=0, world_size=1)
tr = Trainer(device=dev, seed=0, ddp=DDP)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4): tr.step(pool[s % 2])
torch.cuda.synchronize()
import gc, os
if os.environ.get('JITTER_GC', '1') == '0':
    gc.collect(); gc.freeze(); gc.disable()
    print('gc frozen + disabled')
rows = []
_item = torch.Tensor.item
waits = []
def _timed_item(self):
    t = time.perf_counter(); r = _item(self); waits.append(time.perf_counter() - t); return r
torch.Tensor.item = _timed_item
per_step_waits = []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for s in range(40):
    waits.clear()
    t0 = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True)
    loss_dict = tr.wrapped(pool[s % 2])
    losses = torch.stack([v for v in loss_dict.values() if torch.is_tensor(v) and v.requires_grad]).sum()
    t1 = time.perf_counter()
    losses.backward()
    t2 = time.perf_counter()
    tr.optimizer.step()
    t3 = time.perf_counter()
    ev[s + 1].record()
    rows.append((t1 - t0, t2 - t1, t3 - t2))
    per_step_waits.append(list(waits))
torch.cuda.synchronize()
for s, r in enumerate(rows):
    gpu = ev[s].elapsed_time(ev[s + 1])
    flag = " <--" if gpu > 47 or sum(r) * 1e3 > 47 else ""
    w = per_step_waits[s]
    print("step %2d cpu fwd %.1f bwd %.1f opt %.1f = %.1f ms | gpu interval %.1f ms%s | item() waits: %s" % (s, r[0] * 1e3, r[1] * 1e3, r[2] * 1e3, sum(r) * 1e3, gpu, flag, " ".join("%.1f" % (x * 1e3) for x in w)))

import statistics
print("median cpu fwd %.1f bwd %.1f opt %.1f | gpu interval %.1f ms (ddp=%s)" % (
    statistics.median(r[0] for r in rows) * 1e3, statistics.median(r[1] for r in rows) * 1e3,
    statistics.median(r[2] for r in rows) * 1e3, statistics.median(ev[s].elapsed_time(ev[s + 1]) for s in range(40)), DDP))
