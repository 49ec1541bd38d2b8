"""Device time of the phases of one ConQueR training step, backward split at the activations that separate its stages
(the same tensor hooks the bucketed gradient exchange uses, `model.grad_watch`) plus the encoder memory.  GPU box.

    python scripts/ubench/bwd_phases.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0, max_iters=10000)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(4)]
marks = {}


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.setdefault(name, ev)   # the first time the gradient of a watched tensor arrives


def watch(key, tensor):
    if tensor.requires_grad:
        tensor.register_hook(lambda g, key=key: (mark("bwd:" + key), None)[1])


tr.model.grad_watch = watch
enc = tr.model.transformer.encoder


def on_memory(mod, inputs, out):
    if torch.is_tensor(out) and out.requires_grad:
        out.register_hook(lambda g: (mark("bwd:memory"), None)[1])
    return None   # (a forward hook's return value replaces the output)


enc.register_forward_hook(on_memory)
rows = []
for it in range(25):
    marks.clear()
    mark("start")
    tr.optimizer.zero_grad(set_to_none=True)
    loss_dict = tr.wrapped(pool[it % 4])
    total = loss_dict.total() if hasattr(loss_dict, "total") else sum(loss_dict.values())
    mark("fwd_end")
    total.backward()
    mark("bwd_end")
    tr.optimizer.step()
    mark("opt_end")
    torch.cuda.synchronize()
    if it >= 5:
        order = ["start", "fwd_end", "bwd:memory", "bwd:transformer", "bwd:neck", "bwd_end", "opt_end"]
        rows.append([marks[a].elapsed_time(marks[b]) for a, b in zip(order[:-1], order[1:])])
r = np.array(rows).mean(0)
print("device ms between marks (mean of %d steps): forward %.2f | backward: losses + heads + decoder (until the gradient of the "
      "encoder memory is complete) %.2f, encoder %.2f, input projection + neck %.2f, sparse backbone %.2f | optimizer %.2f | sum %.2f"
      % (len(rows), r[0], r[1], r[2], r[3], r[4], r[5], r.sum()))
