"""Device time of selected sub-modules of the step, by CUDA events around their forward (GPU box)."""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
acc = defaultdict(list)


def timed(obj, name, label):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        acc[label].append((e0, e1))
        return out
    setattr(obj, name, wrap)


t = tr.model.transformer
timed(t, "_momentum_update_gt_decoder", "gt momentum update")
timed(t.decoder_gt, "forward", "gt decoder fwd")
timed(t.decoder, "forward", "decoder fwd")
timed(t.encoder, "forward", "encoder fwd")
timed(t, "_get_enc_proposals", "proposals")
timed(tr.model, "_losses", "losses fwd")
timed(tr.model.backbone, "forward", "backbone+fpn fwd")
for s in range(12):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
for k, v in acc.items():
    ms = [a.elapsed_time(b) for a, b in v[4:]]
    print("%-22s %.2f ms" % (k, sum(ms) / len(ms)))
