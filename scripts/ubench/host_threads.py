"""Which threads of a training process burn host CPU in steady state?  (GPU box)  Runs the bench's step loop and
samples per-thread CPU time with psutil; the 8-GPU box shares one CPU quota between 8 such processes."""
import os
import sys
import threading
import time

import psutil
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(10):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
proc = psutil.Process()
names = {t.native_id: t.name for t in threading.enumerate()}
before = {t.id: (t.user_time, t.system_time) for t in proc.threads()}
t0 = time.perf_counter()
n = 300
for s in range(n):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
wall = time.perf_counter() - t0
rows = []
for t in proc.threads():
    b = before.get(t.id, (0.0, 0.0))
    rows.append((t.user_time - b[0], t.system_time - b[1], t.id))
rows.sort(reverse=True)
print("%.2f ms/step; OMP_NUM_THREADS=%s torch threads %d; process threads %d" % (
    wall / n * 1e3, os.environ.get("OMP_NUM_THREADS"), torch.get_num_threads(), len(rows)))
tot = 0.0
for u, s, tid in rows[:10]:
    tot += u + s
    print("  thread %-8d %-18s user %5.1f%%  sys %5.1f%%" % (tid, names.get(tid, "(native)"), 100 * u / wall, 100 * s / wall))
print("sum of all threads: %.2f cores" % (sum(u + s for u, s, _ in rows) / wall))
