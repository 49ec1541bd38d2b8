"""Where do the sporadic slow steps come from?  Host-side phase timestamps of a pipelined run (no per-step sync)."""
import sys, time, torch
sys.path.insert(0, '.')
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device('cuda:0')
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4): tr.step(pool[s % 2])
torch.cuda.synchronize()
rows = []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for s in range(40):
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
torch.cuda.synchronize()
for s, r in enumerate(rows):
    gpu = ev[s].elapsed_time(ev[s + 1])
    flag = " <--" if gpu > 47 or sum(r) * 1e3 > 47 else ""
    print("step %2d cpu fwd %.1f bwd %.1f opt %.1f = %.1f ms | gpu interval %.1f ms%s" % (s, r[0] * 1e3, r[1] * 1e3, r[2] * 1e3, sum(r) * 1e3, gpu, flag))
