"""cProfile of the host side of the training step (GPU box): where do the ~28 ms of issue time per step go?
    python scripts/ubench/host_profile.py [n_steps] [--empty-queues]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
tr = Trainer(device=dev, seed=0, max_iters=10000)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(6):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
if "--empty-queues" in sys.argv:   # the host's OWN time: every step starts from empty device queues (no launch ever blocks)
    for s in range(n):
        torch.cuda.synchronize()
        pr.enable()
        tr.step(pool[s % 2])
        pr.disable()
else:
    pr.enable()
    for s in range(n):
        tr.step(pool[s % 2])
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print("per-step totals over %d steps (tottime / step in ms):" % n)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / n * 1e3, ct / n * 1e3, nc / n, "%s:%d %s" % (fn.replace(ROOT + "/", ""), line, name)))
rows.sort(reverse=True)
for tt, ct, nc, where in rows[:45]:
    print("%7.3f ms self %7.3f ms cum %7.1f calls  %s" % (tt, ct, nc, where[-110:]))
