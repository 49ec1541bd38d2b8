"""List every host-synchronising op of one training step (torch sync debug mode) -- GPU box."""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efg_amd.engine import Trainer, synthetic_batch
dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(3):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
import traceback
torch.cuda.set_sync_debug_mode("warn")
seen = []
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/efg_amd/" in f.filename]
    seen.append((str(message)[:60], ["%s:%d" % (os.path.relpath(f.filename), f.lineno) for f in st[-3:]]))
warnings.showwarning = showwarning
warnings.simplefilter("always")
tr.step(pool[1])
torch.cuda.set_sync_debug_mode("default")
for m, st in seen:
    print(m, " <- ".join(reversed(st)))
print(len(seen), "synchronising calls in one step")
