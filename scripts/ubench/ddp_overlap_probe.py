"""Does the bucketed exchange HIDE wire time on this stack?  One GPU, one rank, a live RCCL communicator -- so there is no wire;
the probe adds it: every gradient all-reduce is followed, on the stream it was issued from, by a spin kernel whose length is
`wire_us` x (elements of the call / all gradient elements) -- what a ring all-reduce of that bucket would occupy the
communication stream for.  flat: the spin sits on the compute stream after backward (exposed by construction); bucket: on
the communication stream beside the rest of backward, only the last bucket's share can be exposed.

    GPU_MAX_HW_QUEUES=2 python scripts/ubench/ddp_overlap_probe.py [wire_us ...]       (default 0 1000 3000)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import configure_hip_runtime  # noqa: E402

configure_hip_runtime()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402

_real_all_reduce = dist.all_reduce
_state = {"wire_us": 0.0, "total": 1, "cycles_per_us": 100.0}


def _all_reduce_with_wire(tensor, *args, **kwargs):
    work = _real_all_reduce(tensor, *args, **kwargs)
    if _state["wire_us"] > 0 and tensor.numel() > 100000:
        torch.cuda._sleep(int(_state["wire_us"] * tensor.numel() / _state["total"] * _state["cycles_per_us"]))
    return work


def calibrate():
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    torch.cuda.synchronize()
    s.record()
    torch.cuda._sleep(10000000)
    e.record()
    torch.cuda.synchronize()
    _state["cycles_per_us"] = 10000000 / (s.elapsed_time(e) * 1e3)


def run(mode, wire_us, steps=20, warmup=6):
    os.environ["EFG_DDP_MODE"] = mode
    _state["wire_us"] = 0.0
    tr = Trainer(device="cuda:0", overrides={"model.transformer.num_queries": 900}, seed=0, ddp=True)
    _state["total"] = sum(p.numel() for p in tr.model.parameters() if p.requires_grad)
    pool = [synthetic_batch(2000 + 100 * p, 2, device="cuda:0") for p in range(2)]
    for w in range(warmup):
        tr.step(pool[w % 2])
    _state["wire_us"] = float(wire_us)
    tr.step(pool[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr.close()
    return 1000 * dt / steps


if __name__ == "__main__":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.all_reduce = _all_reduce_with_wire
    calibrate()
    wires = [float(a) for a in sys.argv[1:]] or [0.0, 1000.0, 3000.0]
    print("GPU_MAX_HW_QUEUES=%s, spin calibration %.1f cycles/us" % (os.environ.get("GPU_MAX_HW_QUEUES"), _state["cycles_per_us"]))
    for wire in wires:
        f = run("flat", wire)
        b = run("bucket", wire)
        print("simulated wire %6.0f us per step: flat %.2f ms/step, bucket %.2f ms/step" % (wire, f, b), flush=True)
    dist.destroy_process_group()
