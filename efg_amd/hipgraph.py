"""Safe HIP-graph capture of a launch-bound, fixed-shape region (PyTorch's CUDAGraph on ROCm = hipGraph).

Why not plain `with torch.cuda.graph(g):`
  * torch >= 2.9 no longer runs the cyclic collector before a capture, and Python may run it DURING one (on any
    thread).  A dead reference cycle that owns an older CUDAGraph then gets destroyed mid-capture; on ROCm
    `~CUDAGraph` calls hipDeviceSynchronize under AT_CUDA_CHECK, which is illegal while a stream of this thread is
    capturing -> the check throws inside a C++ destructor -> std::terminate -> the PROCESS aborts (SIGABRT, no Python
    traceback).  Whether it happens depends on the allocation count of everything that ran before -- the kind of abort
    that shows up on one box and not on another.  Here the collector runs once before the capture and is switched off
    until the capture has ended.
  * if the captured region raises, `torch.cuda.graph.__exit__` calls capture_end(), which raises again and skips the
    stream context's exit: the thread is left on the capture stream for good.  Here the capture is always ended and the
    previous stream restored, and the caller gets the ORIGINAL exception.
"""
import gc

import torch


class CaptureFailed(RuntimeError):
    """The region could not be captured (the original error is the __cause__); nothing is left capturing."""


def capture(fn, device, warmup=2):
    """Run fn() `warmup` times on a side stream (lazy library initialisation, allocator growth), then capture one more
    call into a graph.  Returns (graph, fn's captured result).  Raises CaptureFailed, with the stream state restored,
    if the capture cannot be completed."""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    side = torch.cuda.Stream(device=device)
    gc_was_on = gc.isenabled()
    gc.collect()   # dead cycles (older graphs, tensors) go NOW, not in the middle of the capture
    gc.disable()
    graph = torch.cuda.CUDAGraph()
    try:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize(device)
        side.wait_stream(cur)
        out, failure = None, None
        with torch.cuda.stream(side):
            # thread_local: another thread (data/loader.py) may allocate / launch on its own stream meanwhile
            graph.capture_begin(capture_error_mode="thread_local")
            try:
                out = fn()
            except BaseException as exc:  # noqa: BLE001 -- re-raised below, once the capture has been closed
                failure = exc
            try:
                graph.capture_end()
            except Exception as exc:  # noqa: BLE001 -- an invalidated capture reports itself here
                failure = failure or exc
        cur.wait_stream(side)
        if failure is not None:
            try:
                graph.reset()
            except Exception:  # noqa: BLE001
                pass
            if not isinstance(failure, Exception):
                raise failure   # KeyboardInterrupt / SystemExit: not ours to wrap
            raise CaptureFailed("HIP-graph capture failed: %s" % failure) from failure
        return graph, out
    finally:
        if gc_was_on:
            gc.enable()
