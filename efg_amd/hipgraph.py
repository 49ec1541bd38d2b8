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
  * an INVALIDATED capture (an illegal call inside the region) poisons two things on ROCm that the context manager
    leaves poisoned: the capture stream and the default generator.  See _retire.
"""
import gc

import torch


class CaptureFailed(RuntimeError):
    """The region could not be captured (the original error is the __cause__); nothing is left capturing."""


_capture_streams = {}   # device index -> (raw handle, torch.cuda.ExternalStream); this module's own, never the pool's


def _capture_stream(device):
    """The stream captures on `device` run on: created through the C ABI (efg_capture_stream_create), not taken from
    PyTorch's pool -- see _retire."""
    import ctypes

    from . import _lib

    index = device.index if device.index is not None else torch.cuda.current_device()
    if index not in _capture_streams:
        raw = ctypes.c_void_p()
        with torch.cuda.device(index):
            _lib.check(_lib.lib().efg_capture_stream_create(ctypes.byref(raw)))
        _capture_streams[index] = (raw.value, torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", index)))
    return _capture_streams[index][1]


def _retire(device):
    """After a capture that ROCm invalidated (an illegal call inside the region):

    * the capture stream stays in hipStreamCaptureStatusInvalidated for good -- hipStreamEndCapture reports the error
      and does NOT end the capture (ROCm 7.0 runtime of torch 2.10; scripts/repro/invalidated_stream.py) -- and every
      later launch on it fails.  It is ours, so nothing else ever gets it; destroy it, the next capture makes a new one.
    * torch's capture_end() threw before the generators' epilogue: the device's default generator keeps "capturing"
      set and every later random op raises "Offset increment outside graph capture encountered unexpectedly"; reset(),
      deleting the graph or another capture do not clear it (scripts/repro/failed_capture_variants.py).  Swapping the
      generator's state object for its clone does: seed and offset carry over, the flag does not.  (Graphs captured
      EARLIER stay registered with the old state object and keep drawing from it.)"""
    from . import _lib

    index = device.index if device.index is not None else torch.cuda.current_device()
    raw, _ = _capture_streams.pop(index, (None, None))
    if raw is not None:
        _lib.lib().efg_capture_stream_destroy(raw)   # best effort: a stream that cannot be destroyed is just left alone
    try:
        gen = torch.cuda.default_generators[index]
        gen.graphsafe_set_state(gen.clone_state())
    except Exception:  # noqa: BLE001 -- the original failure is what gets reported
        pass


def capture(fn, device, warmup=2):
    """Run fn() `warmup` times on a side stream (lazy library initialisation, allocator growth), then capture one more
    call into a graph.  Returns (graph, fn's captured result).  Raises CaptureFailed, with the stream state restored,
    if the capture cannot be completed."""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    side = _capture_stream(device)
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
        out, failure, invalidated = None, None, False
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
                failure, invalidated = failure or exc, True
        if failure is not None:
            try:
                graph.reset()
            except Exception:  # noqa: BLE001
                pass
            if invalidated:
                _retire(device)
            else:
                cur.wait_stream(side)
            if not isinstance(failure, Exception):
                raise failure   # KeyboardInterrupt / SystemExit: not ours to wrap
            raise CaptureFailed("HIP-graph capture failed: %s" % failure) from failure
        cur.wait_stream(side)
        return graph, out
    finally:
        if gc_was_on:
            gc.enable()
