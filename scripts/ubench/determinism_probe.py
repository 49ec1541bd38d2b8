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
    events = []   # (phase, module, checksum tensor): the first entry that differs between the runs names the op
    if "--trace" in sys.argv:
        def digest(x):
            ts = []
            def walk(o):
                if torch.is_tensor(o):
                    ts.append(o)
                elif hasattr(o, "features") and torch.is_tensor(getattr(o, "features")):
                    ts.append(o.features)
                elif isinstance(o, (list, tuple)):
                    [walk(i) for i in o]
                elif isinstance(o, dict):
                    [walk(i) for i in o.values()]
            walk(x)
            ts = [t.detach() for t in ts if t.is_cuda and t.numel()]
            if not ts:
                return None
            out = []
            for t in ts:
                w = t.contiguous().view(-1).view(torch.uint8).to(torch.int64)
                out.append((w * (torch.arange(w.numel(), device=w.device) % 251 + 1)).sum())
            return torch.stack(out).sum()
        def on_forward(m, i, o, name):
            events.append(("fwd", name, digest(o)))
            t = o if torch.is_tensor(o) else getattr(o, "features", None)
            if torch.is_tensor(t) and t.requires_grad:
                t.register_hook(lambda g, name=name: events.append(("bwd", name, digest(g))))
        for name, mod in tr.model.named_modules():
            mod.register_forward_hook(lambda m, i, o, name=name: on_forward(m, i, o, name))
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
                 {n: p.detach().clone() for n, p in tr.model.named_parameters()},
                 [(ph, n, None if d is None else int(d)) for ph, n, d in events]))
    tr.close()
    del tr
for s, (a, b) in enumerate(zip(*per_step)):
    print("step %d: total loss %s, sum |grad| %s" % (s, "same" if a[0] == b[0] else "%.9g vs %.9g" % (a[0], b[0]),
                                                     "same" if a[1] == b[1] else "%.12g vs %.12g" % (a[1], b[1])))
(l0, g0, p0, e0), (l1, g1, p1, e1) = runs
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
import ctypes  # noqa: E402

from efg_amd import _lib  # noqa: E402

nfb = ctypes.c_int64(0)
_lib.check(_lib.lib().efg_spconv_streamk_fallbacks(ctypes.byref(nfb), 0))
print("stream-K units recomputed after a missed share (both runs): %d" % nfb.value)
same = [n for n in g0 if torch.equal(g0[n], g1[n])]
print("identical gradients: %d, e.g. %s" % (len(same), same[:5]))
if e0:
    print("traced events: %d vs %d" % (len(e0), len(e1)))
    shown = 0
    for i, (a, b) in enumerate(zip(e0, e1)):
        if a != b:
            print("  event %4d differs: %s %s" % (i, a[0], a[1] if a[1] == b[1] else "%s / %s" % (a[1], b[1])))
            shown += 1
            if shown >= 25:
                break
