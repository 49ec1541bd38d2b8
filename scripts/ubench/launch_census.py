"""Census of one steady-state ConQueR training step (GPU box): host ops by name and thread (forward / autograd), the
kernels each launches, and host self-time -- where the ~2000 launches and the ~30 ms of host work come from.

    python scripts/ubench/launch_census.py [--model conquer]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
if "--model" in sys.argv and sys.argv[sys.argv.index("--model") + 1] == "trajectoryformer":
    import numpy as np

    from efg_amd.tracking import TrajectoryFormer
    from efg_amd.tracking.synthetic import synthetic_tracking_batch

    np.random.seed(1000)
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tr = Trainer(config=os.path.join(root, "configs", "trajectoryformer_waymo_centerpoint.yaml"), device=dev, seed=0,
                 model_cls=TrajectoryFormer, max_iters=1000)
    pool = [synthetic_tracking_batch(7000 + 100 * p, 4, device=dev, n_points=180000, n_objects=60, n_false=20)
            for p in range(2)]
elif "--model" in sys.argv and sys.argv[sys.argv.index("--model") + 1] == "centerpoint":
    from efg_amd.centerpoint import VoxelNet

    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tr = Trainer(config=os.path.join(root, "configs", "centerpoint_waymo_voxelnet.yaml"), device=dev, seed=0,
                 model_cls=VoxelNet, max_iters=1000)
    pool = [synthetic_batch(3000 + 100 * p, 2, device=dev) for p in range(2)]
else:
    tr = Trainer(device=dev, seed=0)
    pool = [synthetic_batch(2000 + 100 * p, 2, device=dev) for p in range(2)]
for s in range(6):
    tr.step(pool[s % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(pool[0])
    torch.cuda.synchronize()
ev = [e for e in prof.events()]
main_thread = ev[0].thread
ops = collections.defaultdict(lambda: [0, 0, 0.0, 0.0])  # count, kernels, self cpu us, kernel us
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    key = (("fwd" if e.thread == main_thread else "bwd"), e.name)
    ops[key][0] += 1
    ops[key][1] += len(e.kernels or [])
    ops[key][2] += e.self_cpu_time_total
    ops[key][3] += sum(k.duration for k in (e.kernels or []))
tot = collections.defaultdict(lambda: [0, 0, 0.0, 0.0])
for (th, name), v in ops.items():
    for i in range(4):
        tot[th][i] += v[i]
for th, v in tot.items():
    print("%s: %d ops, %d kernel launches, self cpu %.2f ms, kernel %.2f ms" % (th, v[0], v[1], v[2] / 1e3, v[3] / 1e3))
print("\n-- by launches")
for (th, name), v in sorted(ops.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%s %-55s %5d ops %5d launches  self cpu %7.2f ms  kernels %7.3f ms" % (th, name[:55], v[0], v[1], v[2] / 1e3, v[3] / 1e3))
print("\n-- by host self time")
for (th, name), v in sorted(ops.items(), key=lambda kv: -kv[1][2])[:35]:
    print("%s %-55s %5d ops %5d launches  self cpu %7.2f ms" % (th, name[:55], v[0], v[1], v[2] / 1e3))
print("\n-- by kernel time")
for (th, name), v in sorted(ops.items(), key=lambda kv: -kv[1][3])[:40]:
    print("%s %-55s %5d ops %5d launches  kernels %7.3f ms" % (th, name[:55], v[0], v[1], v[3] / 1e3))
print("\n-- GEMM / attention shapes by kernel time")
g = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in (
            "aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::_efficient_attention_forward",
            "aten::_efficient_attention_backward", "aten::_flash_attention_forward", "aten::_flash_attention_backward") and e.kernels:
        key = (("fwd" if e.thread == main_thread else "bwd"), e.name, str(e.input_shapes)[:90])
        g[key][0] += 1
        g[key][1] += sum(k.duration for k in e.kernels)
for (th, name, shp), (n, us) in sorted(g.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%s %-12s %3d x %8.1f us total  %s" % (th, name.replace("_efficient_attention", "attn")[:18], n, us, shp))
print("\n-- forward launches by efg:: scope (innermost record_function containing the op)")
scopes = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("efg::") and e.thread == main_thread]
per = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or e.thread != main_thread or not e.kernels or e.name.startswith("efg::"):
        continue
    t0, t1 = e.time_range.start, e.time_range.end
    inner = None
    for sc in scopes:
        if sc.time_range.start <= t0 and t1 <= sc.time_range.end:
            if inner is None or sc.time_range.elapsed_us() < inner.time_range.elapsed_us():
                inner = sc
    key = inner.name if inner is not None else "(none)"
    per[key][0] += len(e.kernels)
    per[key][1] += sum(k.duration for k in e.kernels)
    per[key][2][e.name] += len(e.kernels)
for key, (n, us, ops_) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %5d launches %8.3f ms   top: %s" % (key, n, us / 1e3, ", ".join("%s x%d" % kv for kv in ops_.most_common(6))))
print("\n-- backward launches by autograd node (the enclosing `autograd::engine::evaluate_function: X` range)")
nodes = sorted((e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and e.thread != main_thread
                and e.name.startswith("autograd::engine::evaluate_function")), key=lambda e: e.time_range.start)
starts = [n.time_range.start for n in nodes]
import bisect  # noqa: E402

per_node = collections.defaultdict(lambda: [0, 0, 0.0, collections.Counter()])   # node evaluations, launches, kernel us, ops
for n in nodes:
    per_node[n.name.split(": ", 1)[-1]][0] += 1
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or e.thread == main_thread or not e.kernels:
        continue
    i = bisect.bisect_right(starts, e.time_range.start) - 1
    if i < 0 or e.time_range.end > nodes[i].time_range.end or e.name.startswith("autograd::engine"):
        continue
    rec = per_node[nodes[i].name.split(": ", 1)[-1]]
    rec[1] += len(e.kernels)
    rec[2] += sum(k.duration for k in e.kernels)
    rec[3][e.name] += len(e.kernels)
for name, (n, launches, us, ops_) in sorted(per_node.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-44s %4d nodes %5d launches %8.3f ms   %s" % (name[:44], n, launches, us / 1e3,
                                                         ", ".join("%s x%d" % kv for kv in ops_.most_common(5))))
