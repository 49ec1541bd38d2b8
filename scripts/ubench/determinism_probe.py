"""Which gradients of one ConQueR training step differ between two identical runs?  (GPU box)

Two trainers with the same seed take the same step on the same batch; every parameter's gradient (and the loss terms) is
compared bit for bit.  The parameters that differ name the kernels whose accumulation order is not fixed.

    python scripts/ubench/determinism_probe.py [--steps 1] [--force-proposals]

EFG_DETERMINISTIC=1 (fixed-order weight gradient of the dense 3 x 3 convolution) makes every gradient identical."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
dev = torch.device("cuda:0")
runs = []
seen = []
per_step = []
for r in range(2):
    small = "--small" in sys.argv   # the configuration of tests/ddp_gpu_worker.py
    ov = {"model.transformer.num_queries": 60, "model.transformer.enc_layers": 1, "model.transformer.dec_layers": 2} if small else None
    tr = Trainer(device=dev, seed=0, overrides=ov, max_iters=50 if small else None)
    tr.model.noise_generator = torch.Generator().manual_seed(4321)
    # torch.topk returns ANY members of a tie at the cut (radix select with atomics), and on a random-init model the
    # proposal scores have a plateau of equal values there: the second run takes the first run's proposals, so that what
    # is compared is the kernels' arithmetic, not the tie-break
    tf = tr.model.transformer
    if "--force-proposals" not in sys.argv:
        pass   # csrc/topk.hip breaks ties by index: nothing to force (EFG_TOPK=0 brings torch.topk and its arbitrary ties back)
    elif r == 0:
        real = tf._select_proposals
        tf._select_proposals = lambda probs, real=real: (lambda out: (seen.append(out[1].clone()), out)[1])(real(probs))
    else:
        it = iter(seen)
        tf._select_proposals = lambda probs, it=it: (lambda idx: (torch.gather(probs, 1, idx), idx))(next(it))
    trace = []
    for s in range(steps):
        batch = (synthetic_batch(700 + 10 * s, 1, n_points=30000, n_boxes=12, device=dev) if small else
                 synthetic_batch(2000 + 10 * s, 2, device=dev))
        losses, total = tr.step(batch)
        torch.cuda.synchronize()
        trace.append((float(total.detach()), float(sum(p.grad.double().abs().sum() for p in tr.model.parameters() if p.grad is not None))))
    per_step.append(trace)
    runs.append(({k: float(v.detach()) for k, v in losses.items()},
                 {n: p.grad.detach().clone() for n, p in tr.model.named_parameters() if p.grad is not None},
                 {n: p.detach().clone() for n, p in tr.model.named_parameters()}))
    tr.close()
    del tr
for s, (a, b) in enumerate(zip(*per_step)):
    print("step %d: total loss %s, sum |grad| %s" % (s, "same" if a[0] == b[0] else "%.9g vs %.9g" % (a[0], b[0]),
                                                     "same" if a[1] == b[1] else "%.12g vs %.12g" % (a[1], b[1])))
(l0, g0, p0), (l1, g1, p1) = runs
bad_l = [k for k in l0 if l0[k] != l1[k]]
print("loss terms that differ: %d of %d %s" % (len(bad_l), len(l0), bad_l[:6]))
bad = []
for n in g0:
    if not torch.equal(g0[n], g1[n]):
        d = (g0[n] - g1[n]).abs().max().item() / max(g0[n].abs().max().item(), 1e-30)
        bad.append((d, n))
print("gradients that differ: %d of %d" % (len(bad), len(g0)))
groups = {}
for d, n in bad:
    key = ".".join(n.split(".")[:4])
    groups.setdefault(key, [0, 0.0])
    groups[key][0] += 1
    groups[key][1] = max(groups[key][1], d)
for k, (c, d) in sorted(groups.items()):
    print("  %-60s %3d tensors, max rel diff %.2e" % (k, c, d))
same = [n for n in g0 if torch.equal(g0[n], g1[n])]
print("identical gradients: %d, e.g. %s" % (len(same), same[:5]))
