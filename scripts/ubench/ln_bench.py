import sys, torch
sys.path.insert(0, '.')
from efg_amd.operators.layernorm import add_layer_norm
from scripts.bench_ops import timeit
dev='cuda'
x=torch.randn(2,70688,256,device=dev,requires_grad=True); r=torch.randn(2,70688,256,device=dev,requires_grad=True)
norm=torch.nn.LayerNorm(256).to(dev); dy=torch.randn_like(x)
for name,fn in (("torch", lambda: norm(x+r)), ("fused", lambda: add_layer_norm(x,r,norm))):
    tf=timeit(fn)
    y=fn()
    tb=timeit(lambda: torch.autograd.grad(y,(x,r,norm.weight,norm.bias),dy,retain_graph=True))
    print(name,"fwd %.1f us  bwd %.1f us"%(tf,tb))
