"""Which parameter gradients reach AccumulateGrad as views or with strides other than the parameter's?  (The latter cost a copy
launch each: the transposed weight gradients of the decoder-sized Linear layers and the channels-last FPN convolution, ~25 per step,
~0.1 ms.)  GPU box.
    python scripts/ubench/acc_probe.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from efg_amd.engine import Trainer, synthetic_batch, configure_hip_runtime
configure_hip_runtime()
dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(3):
    tr.step(pool[s % 2])
bad = []
hs = []
for n, p in tr.model.named_parameters():
    if not p.requires_grad: continue
    def mk(n, p):
        def h(g):
            if g.stride() != p.stride() or g._base is not None and not g.is_contiguous():
                bad.append((n, tuple(g.shape), g.stride(), p.stride(), g._base is not None))
            elif g._base is not None:
                bad.append((n + " (view, contiguous)", tuple(g.shape), g.stride(), p.stride(), True))
        return h
    hs.append(p.register_hook(mk(n, p)))
tr.step(pool[1])
torch.cuda.synchronize()
print(len(bad))
for b in bad: print(b)
