"""Tune the dense fp32 GEMMs of the train step with PyTorch TunableOp (hipBLASLt / rocBLAS solution search).

Run on the GPU box:  EFG_TUNE_ROTATE_MB=1024 EFG_TUNED_GEMMS=0 python scripts/tune_gemm.py [out.csv]
(the committed efg_amd/tuned/gemm_gfx950.csv was produced this way: operands rotate through 1 GB so every
candidate is timed with cold caches, which is how the 72 MB activations reach these GEMMs inside the step)
Runs a few train steps with tuning enabled (every new GEMM shape is benchmarked over the library's solutions
once), writes the chosen solutions to a CSV, then reports the step time with tuning frozen.  The CSV is plumbing
(library algorithm selection) -- it changes no arithmetic type; entries are validated against the ROCm /
hipBLASLt versions at load time and ignored when they do not match."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
out = os.path.abspath(args[0] if args else "gpurun_out/tunableop_results.csv")
# --queries N: tune the shapes of another query count (BASELINE configs[2] names 900; the reference YAML 1000); merge the
# new rows into efg_amd/tuned/gemm_gfx950.csv by hand (same validators)
queries = int(sys.argv[sys.argv.index("--queries") + 1]) if "--queries" in sys.argv else None
os.makedirs(os.path.dirname(out), exist_ok=True)

import torch.cuda.tunable as tunable
from efg_amd.engine import Trainer, synthetic_batch

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0, overrides={"model.transformer.num_queries": queries} if queries else None)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]


def steps(n, tag):
    ts = []
    for s in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        tr.step(pool[s % 2])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    print(tag, " ".join("%.1f" % x for x in ts), flush=True)
    return ts


tunable.enable(False)
steps(4, "untuned:")
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(20)
if os.environ.get("EFG_TUNE_ROTATE_MB"):  # cold-cache timing: rotate operands through a buffer larger than L2 + MALL
    tunable.set_rotating_buffer_size(int(os.environ["EFG_TUNE_ROTATE_MB"]))
t0 = time.perf_counter()
steps(3, "tuning:")
print("tuning took %.1f s, %d entries" % (time.perf_counter() - t0, len(tunable.get_results())), flush=True)
# (the results file is written as tuning proceeds)
tunable.tuning_enable(False)
steps(8, "tuned:")
print("validators:", tunable.get_validators())
