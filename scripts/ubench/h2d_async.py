"""Does a small host->device upload block the host while the stream is busy?  (GPU box)"""
import time
import torch

dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev)


def busy():
    for _ in range(20):
        a @ a  # ~20 x 7 ms


def probe(name, fn):
    torch.cuda.synchronize()
    busy()
    t = time.perf_counter()
    fn()
    dt = (time.perf_counter() - t) * 1e3
    torch.cuda.synchronize()
    print("%-34s host blocked %.2f ms" % (name, dt))


h = torch.randn(80, 7)
hp = torch.randn(80, 7).pin_memory()
for _ in range(2):
    probe("nothing", lambda: None)
    probe("pageable .to(non_blocking)", lambda: h.to(dev, non_blocking=True))
    probe("pinned .to(non_blocking)", lambda: hp.to(dev, non_blocking=True))
    probe("pin_memory().to(non_blocking)", lambda: h.pin_memory().to(dev, non_blocking=True))
    probe("torch.tensor(list, device)", lambda: torch.tensor([[188, 188]], dtype=torch.int64, device=dev))
    probe("torch.zeros(device)", lambda: torch.zeros(80, 7, device=dev))
    probe("torch.arange(device)", lambda: torch.arange(100, device=dev))
    s2 = torch.cuda.Stream(priority=-1)
    def side():
        with torch.cuda.stream(s2):
            x = torch.zeros(1, dtype=torch.int32, device=dev) + 3
            return x.item()
    probe("side-stream kernel + .item()", side)
