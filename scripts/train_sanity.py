"""End-to-end sanity: train for N steps on a fixed pool of synthetic batches and print the loss trajectory
(GPU box).  Run twice (e.g. with EFG_FUSED_LN=0 EFG_FUSED_BN=0) to compare the fused kernels against PyTorch's."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efg_amd.engine import Trainer, synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(3000 + 10 * p, 2, device=dev) for p in range(4)]
t0 = time.perf_counter()
hist = []
for s in range(n):
    loss_dict, total = tr.step(pool[s % len(pool)])
    if s % 10 == 0 or s == n - 1:
        hist.append((s, float(total), float(loss_dict["loss_ce"]), float(loss_dict["loss_bbox"]), float(loss_dict["loss_giou"])))
torch.cuda.synchronize()
for h in hist:
    print("step %4d total %.4f  ce %.4f bbox %.4f giou %.4f" % h)
print("finite:", all(torch.isfinite(p).all().item() for p in tr.model.parameters()), " %.1f s" % (time.perf_counter() - t0))
