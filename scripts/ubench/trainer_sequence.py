"""Does a training step get slower for the 2nd / 3rd Trainer built in one process?  (GPU box)
    python scripts/ubench/trainer_sequence.py [--gc] [--keep]   (--gc: gc.collect() between trainers; --keep: no empty_cache())"""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for n in range(4):
    tr = Trainer(device=dev, seed=0, max_iters=10000)
    for s in range(5):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(15):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 15 * 1e3
    print("trainer %d: %.2f ms/step; gc enabled %s, frozen %d, reserved %.1f GB, allocated %.1f GB" % (
        n, dt, gc.isenabled(), gc.get_freeze_count(), torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30))
    tr.close()
    del tr
    if "--gc" in sys.argv:
        gc.collect()
    if "--keep" not in sys.argv:   # --keep: the next trainer reuses the allocator's cached blocks (same physical memory)
        torch.cuda.empty_cache()
