"""Live kernel timing for bench.py: HIP events (on the stream the kernels are launched on --
torch's current stream) around the calls of ONE designated kernel family, with the algorithmic
bytes / flops the caller attributes to each launch."""
import torch

_enabled = False
_records = {}   # name -> list of (start_event, end_event, bytes, flops)
TARGETS = None  # set of names to time; None = all instrumented calls


def enable(flag):
    global _enabled
    _enabled = bool(flag)
    if flag:
        _records.clear()


def active(name):
    return _enabled and (TARGETS is None or name in TARGETS)


class timed:
    """with timed("conv_fwd", bytes, flops): <launch>"""

    def __init__(self, name, nbytes=0, flops=0):
        self.name, self.nbytes, self.flops = name, nbytes, flops
        self.on = active(name)

    def __enter__(self):
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            _records.setdefault(self.name, []).append((self.s, self.e, self.nbytes, self.flops))
        return False


def summary():
    out = {}
    for name, recs in _records.items():
        ms = [s.elapsed_time(e) for s, e, _, _ in recs]
        out[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_us": 1000.0 * sum(ms) / max(len(ms), 1),
                     "bytes": sum(r[2] for r in recs), "flops": sum(r[3] for r in recs)}
    return out


def roofline():
    return None
