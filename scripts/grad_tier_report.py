"""The gradient tier of the full-model golden as a MEASURED table (review item: "the 3e-3 tier should be a table, not a tolerance").

For every gradient tensor the golden stores (tests/golden/conquer_full_small.npz: the reference's fp32 CPU run) and its fp64 twin
(tests/golden/conquer_full_small_grad64.npz: the SAME reference model run in float64 by scripts/make_golden_full.py --grad64-out):
    ref32-vs-64   how far the reference's own fp32 gradient is from its fp64 run        (the conditioning of the quantity)
    ours-vs-64    how far this build's gradient (HIP ops on the GPU, oracle ops on the CPU) is from the fp64 run
    ours-vs-ref32 what tests/test_model_full_golden.py bounds
all as max |difference| / max |fp64 gradient|.  Run with EFG_DETERMINISTIC=1 on the GPU for a run-to-run stable table.
    EFG_DETERMINISTIC=1 python scripts/grad_tier_report.py > profiles/r06_grad_tier_table.txt"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_full_golden as T  # noqa: E402

dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
model, g = T._build(dev, full_graph=True)
g64 = np.load(os.path.join(ROOT, "tests", "golden", "conquer_full_small_grad64.npz"))
if dev.type == "cpu":
    from oracle import cpu_backend

    ctx = cpu_backend.install()
else:
    ctx = contextlib.nullcontext()
with ctx:
    cap, losses, total = T._run(model, dev)
params = dict(model.named_parameters())
print("# device %s, EFG_DETERMINISTIC=%s; total loss ours %.9f, reference fp32 %.9f, reference fp64 %.9f" % (
    dev, os.environ.get("EFG_DETERMINISTIC", "0"), float(total), float(g["total_loss"]), float(g64["total_loss64"])))
print("# %-72s %-18s %10s %12s %12s %13s  %s" % ("gradient", "shape", "max|g64|", "ref32-vs-64", "ours-vs-64", "ours-vs-ref32", "tier"))
rows = []
for k in sorted(g):
    if not k.startswith("grad::"):
        continue
    name = k[6:]
    got = params[name].grad.detach().double().cpu().numpy()
    if got.size > 65536:
        got = got[:8]
    ref32, ref64 = g[k].astype(np.float64), g64["grad64::" + name]
    scale = np.abs(ref64).max()
    e_ref, e_ours, e_32 = np.abs(ref32 - ref64).max() / scale, np.abs(got - ref64).max() / scale, np.abs(got - ref32).max() / scale
    tier = "exact (1e-4)" if any(e in k for e in T.EXACT_GRADS) else "behind a BatchNorm / the bilinear kink"
    rows.append((name, got.shape, scale, e_ref, e_ours, e_32, tier))
    print("  %-72s %-18s %10.3e %12.2e %12.2e %13.2e  %s" % (name, "x".join(map(str, got.shape)), scale, e_ref, e_ours, e_32, tier))
worst = max(rows, key=lambda r: r[4])
ratio = max(r[4] / max(r[3], 1e-7) for r in rows)
print("# worst ours-vs-64: %.2e (%s); largest ratio ours-vs-64 / ref32-vs-64 over the table: %.2f" % (worst[4], worst[0], ratio))
