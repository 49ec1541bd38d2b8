"""Minimal repro of the "a collective beside backward doubles the step" pathology (profiles/r04_ddp_modes_hw_queues.txt),
outside the trainer and without any communicator.

Three roles, as in a training step:
  main     a long chain of kernels (the backward pass); an event is recorded half way through it
  waiter   one or more side streams that WAIT for that event and then run something small (the bucketed exchange's
           collectives, the matching stream of round 4)
  reader   a high-priority stream that runs a small kernel whose result the HOST reads back at once (the geometry
           stream's voxel / site counts of the next step)
Measured: how long the host sits in the reader's read-back.  It should be microseconds -- the reader's kernel depends on
nothing.  If the runtime has put the reader's stream on the hardware queue that also carries a waiter's barrier packet,
the read-back returns only when the main stream reaches the event.

One process per setting (GPU_MAX_HW_QUEUES is read when the runtime initialises):
    for q in 2 4 8; do for w in 0 1 2 3; do GPU_MAX_HW_QUEUES=$q python scripts/repro/hw_queue_stall.py --waiters $w; done; done
"""
import argparse
import os
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--waiters", type=int, default=1)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reader-priority", type=int, default=-1)
ap.add_argument("--create-first", default="reader", choices=["reader", "waiters"], help="which side streams are created first")
args = ap.parse_args()
dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
small = torch.zeros(1024, device=dev)
main = torch.cuda.current_stream(dev)
if args.create_first == "reader":
    reader = torch.cuda.Stream(dev, priority=args.reader_priority)
    waiters = [torch.cuda.Stream(dev) for _ in range(args.waiters)]
else:
    waiters = [torch.cuda.Stream(dev) for _ in range(args.waiters)]
    reader = torch.cuda.Stream(dev, priority=args.reader_priority)
for _ in range(3):
    torch.mm(a, b)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    torch.mm(a, b)
torch.cuda.synchronize()
mm_ms = (time.perf_counter() - t) * 100.0   # per product
half = max(1, int(round(15.0 / mm_ms)))       # ~15 ms of products on either side of the event

blocked, step_ms = [], []
for s in range(args.steps):
    t0 = time.perf_counter()
    for _ in range(half):
        torch.mm(a, b)
    ev = torch.cuda.Event()
    ev.record(main)
    for _ in range(half):
        torch.mm(a, b)
    for w in waiters:
        w.wait_event(ev)
        with torch.cuda.stream(w):
            small.add_(1.0)
    with torch.cuda.stream(reader):
        r = small.new_ones(8).sum()
        t1 = time.perf_counter()
        float(r)                      # the read-back
        blocked.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    step_ms.append((time.perf_counter() - t0) * 1e3)
blocked, step_ms = blocked[3:], step_ms[3:]
print("GPU_MAX_HW_QUEUES=%s waiters=%d created_first=%s: main chain %.1f ms per step; host blocked in the reader's read-back "
      "%.2f ms (median), %.2f max; step %.1f ms" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), args.waiters, args.create_first,
                                                    2 * half * mm_ms, sorted(blocked)[len(blocked) // 2], max(blocked),
                                                    sum(step_ms) / len(step_ms)))
