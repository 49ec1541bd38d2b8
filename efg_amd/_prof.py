"""Live kernel timing for bench.py: HIP events (on the stream the kernels are launched on --
torch's current stream) around the launches of our HIP kernels, keyed by kernel symbol, with the
ALGORITHMIC bytes / flops the caller attributes to each launch.  Off by default:
no events are created unless bench.py enables it."""
import torch

_enabled = False
_records = {}   # kernel symbol -> list of (start_event, end_event, bytes, flops)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); ~6300 measured achievable
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak


def enable(flag):
    global _enabled
    _enabled = bool(flag)
    if flag:
        _records.clear()


def active():
    return _enabled


class timed:
    """with timed("conv_fwd_kernel<4>", lambda: (bytes, flops)): <single kernel launch>"""

    def __init__(self, name, cost):
        self.name, self.cost = name, cost
        self.on = _enabled

    def __enter__(self):
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            _records.setdefault(self.name, []).append((self.s, self.e, self.cost))  # cost evaluated after the run
        return False


def summary():
    out = {}
    for name, recs in _records.items():
        ms = [s.elapsed_time(e) for s, e, _ in recs]
        costs = [c() for _, _, c in recs]
        out[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_us": 1000.0 * sum(ms) / max(len(ms), 1),
                     "bytes": sum(c[0] for c in costs), "flops": sum(c[1] for c in costs)}
    return out


def roofline(traffic_bytes_per_launch=None):
    """The roofline object of bench.py for the DOMINANT kernel (largest total time among ours)."""
    summ = summary()
    if not summ:
        return None
    name = max(summ, key=lambda k: summ[k]["total_ms"])
    v = summ[name]
    secs = v["total_ms"] / 1e3
    gbs = v["bytes"] / secs / 1e9
    tfs = v["flops"] / secs / 1e12
    # bound: the sparse-conv implicit GEMMs run on the f32 MFMA pipe (whichever roof they sit closer to);
    # every other kernel of the path is gather / scatter / scan work: HBM
    if name.startswith("conv_") and tfs / F32_MFMA_PEAK_TFLOPS > gbs / HBM_PEAK_GBS:
        return {"kernel": name, "bound": "mfma", "achieved": tfs, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tfs / F32_MFMA_PEAK_TFLOPS, "traffic": traffic_bytes_per_launch,
                "avg_launch_us": v["avg_us"], "launches": v["launches"],
                "alg_bytes_per_launch": v["bytes"] / v["launches"], "alg_flops_per_launch": v["flops"] / v["launches"],
                "hbm_GBps_algorithmic": gbs}
    return {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": traffic_bytes_per_launch, "avg_launch_us": v["avg_us"],
            "launches": v["launches"], "alg_bytes_per_launch": v["bytes"] / v["launches"],
            "alg_flops_per_launch": v["flops"] / v["launches"], "TFLOPs_algorithmic": tfs}
