"""Which source lines of efg_amd/ issue the small ATen ops of one training step?  (GPU box)

A TorchDispatchMode records, for every aten op of the FORWARD pass, the innermost Python frame inside efg_amd/ -- the
backward launches of an op (SelectBackward0, ClampBackward1, AddmmBackward0 ...) belong to the line that issued the
forward op, so the forward census locates both.  Ops on tensors with >= --big elements are listed separately.

    python scripts/ubench/op_sites.py [--top 70] [--big 2000000]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=70)
ap.add_argument("--big", type=int, default=2000000)
args = ap.parse_args()

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(4):
    tr.step(pool[s % 2])
torch.cuda.synchronize()

SKIP = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.reshape",
        "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.unbind", "aten.split",
        "aten.sym_", "aten.is_", "aten.size", "aten.stride", "aten.lift_fresh", "aten.empty", "aten._to_copy",
        "aten.slice", "aten.select")   # views launch nothing forward; select / slice are counted (their BACKWARD launches)
COUNT_VIEWS = ("aten.slice", "aten.select")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(name.startswith(s) for s in SKIP) and not any(name.startswith(s) for s in COUNT_VIEWS):
            return out
        numel = 0
        for a in list(args) + ([out] if torch.is_tensor(out) else []):
            if torch.is_tensor(a):
                numel = max(numel, a.numel())
        needs_grad = any(torch.is_tensor(a) and a.requires_grad for a in args)
        if any(name.startswith(s) for s in COUNT_VIEWS) and not needs_grad:
            return out   # a view of a tensor without gradient launches nothing, forward or backward
        site = None
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "/efg_amd/" in fr.filename and "_prof.py" not in fr.filename:
                site = "%s:%d %s" % (fr.filename.split("/efg_amd/")[1], fr.lineno, fr.name)
                break
        self.sites[site or "(outside efg_amd)"][(name.replace("aten.", "").replace(".default", ""),
                                                 "big" if numel >= args_big else "small", needs_grad)] += 1
        return out


args_big = args.big
c = Census()
with c:
    loss_dict = tr.wrapped(pool[0])
    losses = torch.stack([v for v in loss_dict.values() if torch.is_tensor(v) and v.requires_grad]).sum()
losses.backward()
torch.cuda.synchronize()
rows = []
for site, ctr in c.sites.items():
    small = sum(n for (op, sz, g), n in ctr.items() if sz == "small")
    grad = sum(n for (op, sz, g), n in ctr.items() if sz == "small" and g)
    rows.append((small, grad, site, ctr))
rows.sort(key=lambda r: -r[0])
print("forward aten ops on small tensors: %d (of which %d carry a gradient => backward launches too); sites: %d" % (
    sum(r[0] for r in rows), sum(r[1] for r in rows), len(rows)))
for small, grad, site, ctr in rows[: args.top]:
    ops = ", ".join("%s%s x%d" % (op, "*" if g else "", n) for (op, sz, g), n in ctr.most_common(7) if sz == "small")
    print("%4d ops (%3d with grad)  %-58s %s" % (small, grad, site[:58], ops[:150]))
