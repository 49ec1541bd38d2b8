"""GPU box: one full-size training step on the HIP path and on the CPU oracle path from the same weights / inputs;
per parameter GROUP, || g_hip - g_oracle || / || g_oracle || and the ten worst parameters.  Localises a gradient
discrepancy the loss comparison does not show.
    python scripts/parity_groups.py [--model conquer|voxeldetr] [--scenes N] [--queries Q] [--points P]"""
import argparse
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402
from oracle import cpu_backend  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="conquer")
ap.add_argument("--scenes", type=int, default=2)
ap.add_argument("--queries", type=int, default=1000)
ap.add_argument("--points", type=int, default=180000)
args = ap.parse_args()
cfg = None if args.model == "conquer" else os.path.join(ROOT, "configs", "voxeldetr_waymo_res18.yaml")


def run(device, install):
    np.random.seed(3)
    tr = Trainer(config=cfg, device=device, seed=0, ddp=False, overrides={"model.transformer.num_queries": args.queries})
    tr.model.noise_generator = torch.Generator().manual_seed(4321)
    keep = {}

    def hook(mod, inp, out):   # Transformer.forward: which tokens became queries, and their scores
        keep["topk"] = mod.enc_outputs["topk_indexes"].detach().cpu()[..., 0]
        keep["logits"] = mod.enc_outputs["pred_logits"].detach().cpu()[..., 0]

    tr.model.transformer.register_forward_hook(hook)
    with install():
        loss_dict, total = tr.step(synthetic_batch(1000, args.scenes, n_points=args.points,
                                                   device=device if device.type == "cuda" else None))
    run.keep.append(keep)
    grads = {n: p.grad.detach().double().cpu() for n, p in tr.model.named_parameters() if p.grad is not None}
    losses = {k: float(v.detach()) for k, v in loss_dict.items()}
    tr.close()
    return losses, grads


run.keep = []
torch.set_num_threads(16)
lc, gc_ = run(torch.device("cpu"), cpu_backend.install)
lg, gg = run(torch.device("cuda:0"), contextlib.nullcontext)
kc, kg = run.keep
for b in range(kc["topk"].shape[0]):
    sc, sg = set(kc["topk"][b].tolist()), set(kg["topk"][b].tolist())
    only_c, only_g = sorted(sc - sg), sorted(sg - sc)
    print("scene %d: proposal tokens only on cpu %s / only on gpu %s" % (b, only_c[:8], only_g[:8]))
    for t in only_c[:4] + only_g[:4]:
        kth = float(kc["logits"][b][kc["topk"][b]].min())
        print("   token %d: logit cpu %.7f gpu %.7f   (k-th best logit on cpu %.7f)" % (t, float(kc["logits"][b, t]), float(kg["logits"][b, t]), kth))
for k in sorted(lc, key=lambda k: -abs(lg[k] - lc[k]) / max(abs(lc[k]), 1e-6))[:6]:
    print("   loss %-28s cpu %.7f gpu %.7f rel %.2e" % (k, lc[k], lg[k], abs(lg[k] - lc[k]) / max(abs(lc[k]), 1e-6)))
print("losses: max rel diff %.2e" % max(abs(lg[k] - lc[k]) / max(abs(lc[k]), 1e-6) for k in lc))
nc = float(torch.sqrt(sum((g ** 2).sum() for g in gc_.values())))
ng = float(torch.sqrt(sum((g ** 2).sum() for g in gg.values())))
print("grad norm cpu %.6f gpu %.6f rel %.2e" % (nc, ng, abs(ng - nc) / nc))
groups = {}
for n in gc_:
    key = ".".join(n.split(".")[:3])
    d, r = groups.setdefault(key, [0.0, 0.0])
    groups[key] = [d + float(((gg[n] - gc_[n]) ** 2).sum()), r + float((gc_[n] ** 2).sum())]
for key, (d, r) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print("%-60s |diff| %.3e  |ref| %.3e  rel %.2e" % (key, d ** 0.5, r ** 0.5, (d / max(r, 1e-30)) ** 0.5))
worst = sorted(gc_, key=lambda n: -float(((gg[n] - gc_[n]) ** 2).sum()))[:10]
for n in worst:
    print("  %-70s rel %.2e  |ref| %.3e" % (n, float((gg[n] - gc_[n]).norm() / gc_[n].norm().clamp_min(1e-30)), float(gc_[n].norm())))
