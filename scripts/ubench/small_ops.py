"""Where do the small library launches of one training step come from?  (GPU box)

For every kernel of one steady-state ConQueR step that is NOT one of ours (ATen elementwise / reduce / fill / copy,
rocclr memset / copy): the aten op, its input shapes, forward or autograd thread, and the innermost efg_amd source
line on the Python stack -- sorted by launches and by kernel time.

    python scripts/ubench/small_ops.py [--all]        (--all: our kernels and the GEMMs too)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Trainer(device=dev, seed=0)
pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(6):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(pool[0])
    torch.cuda.synchronize()
events = list(prof.events())
main_thread = events[0].thread
show_all = "--all" in sys.argv


def ours(kname):
    return "efg::" in kname or kname.startswith("Cijk") or "igemm" in kname or "ck::" in kname or "_ZN2ck" in kname


def site(e):
    for fr in (e.stack or []):
        if "efg_amd" in fr and "site-packages" not in fr:
            return fr.split("efg_amd/")[-1][:70]
    return "-"


agg = collections.defaultdict(lambda: [0, 0.0])
tot_n, tot_us = 0, 0.0
for e in events:
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    for k in (e.kernels or []):
        if not show_all and ours(k.name):
            continue
        key = ("fwd" if e.thread == main_thread else "bwd", e.name, str(e.input_shapes)[:90], site(e), k.name[:60])
        agg[key][0] += 1
        agg[key][1] += k.duration
        tot_n += 1
        tot_us += k.duration
print("library small kernels in one step: %d launches, %.2f ms" % (tot_n, tot_us / 1e3))
print("\n-- by kernel time")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%4d x %8.1f us  %s %-28s %-90s %s | %s" % (n, us, key[0], key[1][:28], key[2], key[3], key[4]))
by_site = collections.defaultdict(lambda: [0, 0.0])
for key, (n, us) in agg.items():
    by_site[(key[0], key[3])][0] += n
    by_site[(key[0], key[3])][1] += us
print("\n-- by source line")
for key, (n, us) in sorted(by_site.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%4d x %8.1f us  %s %s" % (n, us, key[0], key[1]))
