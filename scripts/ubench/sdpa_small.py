"""fp32 attention over many short sequences (TrajectoryFormer's point encoder: 1232 x 4 heads x 128 tokens x 64):
library SDPA backends vs the explicit bmm / softmax composite.  GPU box."""
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

dev = torch.device("cuda:0")
b, h, l, d = 1232, 4, 128, 64
q, k, v = (torch.randn(b, h, l, d, device=dev, requires_grad=True) for _ in range(3))
go = torch.randn(b, h, l, d, device=dev)


def explicit(q, k, v):
    p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    return torch.matmul(p, v)


def run(fn, tag):
    for _ in range(3):
        out = fn(q, k, v)
        out.backward(go)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(10):
        out = fn(q, k, v)
    e[1].record()
    for _ in range(10):
        out = fn(q, k, v)
        out.backward(go)
    e[2].record()
    torch.cuda.synchronize()
    f = e[0].elapsed_time(e[1]) / 10
    print("%-28s fwd %.3f ms  fwd+bwd %.3f ms" % (tag, f, e[1].elapsed_time(e[2]) / 10))
    return out


ref = run(lambda q, k, v: F.scaled_dot_product_attention(q, k, v), "sdpa default")
for name, be in (("sdpa MATH", SDPBackend.MATH), ("sdpa EFFICIENT", SDPBackend.EFFICIENT_ATTENTION),
                 ("sdpa FLASH", SDPBackend.FLASH_ATTENTION)):
    try:
        def f(q, k, v, be=be):
            with sdpa_kernel(be):
                return F.scaled_dot_product_attention(q, k, v)
        run(f, name)
    except Exception as ex:  # noqa: BLE001
        print(name, "unavailable:", str(ex)[:80])
out = run(explicit, "explicit bmm+softmax")
print("max |explicit - sdpa|", float((out - ref).abs().max()))
