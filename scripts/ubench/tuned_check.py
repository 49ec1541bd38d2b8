"""Is efg_amd/tuned/gemm_gfx950.csv actually applied?  (GPU box)  Prints the number of TunableOp results loaded after a
few steps, the validators, and the step time with and without the file."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.cuda.tunable as tunable  # noqa: E402
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(6):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
print("tunable enabled", tunable.is_enabled(), "tuning", tunable.tuning_is_enabled(), "file", tunable.get_filename())
res = tunable.get_results()
print("results loaded:", len(res), "validators:", tunable.get_validators())


def ms(n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s in range(n):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("with the file: %.2f ms/step" % ms())
tunable.enable(False)
print("TunableOp off: %.2f ms/step" % ms())
tunable.enable(True)
print("with the file: %.2f ms/step" % ms())
