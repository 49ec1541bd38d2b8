"""After a capture that fails, is the default CUDA generator still usable?  (GPU box)"""
import torch
from efg_amd.hipgraph import capture, CaptureFailed

x = torch.zeros(1024, device="cuda")
calls = []


def region():
    calls.append(1)
    if len(calls) == 3:
        return float(x.sum())   # illegal inside a capture
    return x + 1


try:
    capture(region, "cuda:0")
except CaptureFailed as exc:
    print("capture failed as expected:", str(exc)[:80])
try:
    r = torch.rand(4, device="cuda")
    torch.cuda.synchronize()
    print("RNG after failed capture OK", r.sum().item() >= 0)
except Exception as exc:  # noqa: BLE001
    print("RNG after failed capture BROKEN:", str(exc)[:120])
