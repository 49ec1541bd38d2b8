"""cProfile of the main (forward-issuing) thread over a few training steps: where does the host time go?  GPU box."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

tr = Trainer(device="cuda:0", seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device="cuda:0") for p in range(2)]
for s in range(6):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
N = 10
pr = cProfile.Profile()
pr.enable()
for s in range(N):
    tr.step(pool[s % 2])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / N * 1e3, ct / N * 1e3, nc // N, "%s:%d %s" % (os.path.relpath(fn) if fn.startswith("/") else fn, line, name)))
rows.sort(reverse=True)
print("tottime ms/step | cumtime ms/step | calls/step | function")
for r in rows[: int(sys.argv[1]) if len(sys.argv) > 1 else 45]:
    print("%7.2f %8.2f %6d  %s" % r)
