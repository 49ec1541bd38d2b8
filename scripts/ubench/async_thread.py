"""What wakes ROCr's async-event thread?  Tiny-kernel launch loops with and without events / allocations, sampling the
per-thread CPU time of this process.  GPU box."""
import sys
import time

import psutil
import torch

dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
proc = psutil.Process()


def sample(fn, n, tag):
    torch.cuda.synchronize()
    before = {t.id: t.user_time + t.system_time for t in proc.threads()}
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    use = sorted(((t.user_time + t.system_time - before.get(t.id, 0.0)) / wall for t in proc.threads()), reverse=True)
    print("%-46s %7.1f us/iter  threads: %s" % (tag, wall / n * 1e6, " ".join("%.0f%%" % (100 * u) for u in use[:4])))


def launches(n):
    for _ in range(n):
        x.add_(1.0)


def launches_alloc(n):
    for _ in range(n):
        y = x + 1.0  # noqa: F841  (allocation + free through the caching allocator)


def launches_events(n):
    e = torch.cuda.Event()
    for i in range(n):
        x.add_(1.0)
        if i % 10 == 0:
            e.record()


def launches_two_streams(n):
    s = torch.cuda.Stream()
    for i in range(n):
        x.add_(1.0)
        if i % 50 == 0:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                x.mul_(1.0)
            torch.cuda.current_stream().wait_stream(s)


def big_kernels(n):
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(n // 100):
        a @ a


for fn, n, tag in ((launches, 200000, "tiny in-place kernels"), (launches_alloc, 200000, "tiny kernels + allocator"),
                   (launches_events, 200000, "tiny kernels + an event every 10"),
                   (launches_two_streams, 200000, "tiny kernels + a cross-stream wait every 50"),
                   (big_kernels, 20000, "large GEMMs (few launches)")):
    sample(fn, n, tag)
