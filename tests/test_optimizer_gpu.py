"""CachedFusedAdamW against torch.optim.AdamW(fused=True): bit-identical parameters and state."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models():
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 8)).cuda()
    b = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 8)).cuda()
    b.load_state_dict(a.state_dict())
    return a, b


def _groups(m):
    return [{"params": list(m[0].parameters()), "lr": 1e-3}, {"params": list(m[2].parameters())},
            {"params": list(m[3].parameters()), "lr": 5e-3}]  # m[3] never gets a gradient (an unused branch)


def test_cached_fused_adamw_matches_torch():
    from efg_amd.detection3d.optimizer import CachedFusedAdamW

    a, b = _models()
    kw = dict(lr=2e-3, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8)
    ref = torch.optim.AdamW(_groups(a), fused=True, **kw)
    mine = CachedFusedAdamW(_groups(b), **kw)
    g = torch.Generator(device="cuda").manual_seed(1)
    for it in range(5):
        x = torch.randn(64, 16, device="cuda", generator=g)
        for m, opt in ((a, ref), (b, mine)):
            opt.zero_grad(set_to_none=True)
            m[2](m[1](m[0](x))).square().mean().backward()
            if it == 3:  # a scheduler changed the learning rates
                for grp in opt.param_groups:
                    grp["lr"] *= 0.5
            opt.step()
        for p, q in zip(a.parameters(), b.parameters()):
            assert torch.equal(p, q), it
    sa, sb = ref.state_dict(), mine.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        for name in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(sa["state"][k][name], sb["state"][k][name]), (k, name)
    # a parameter set that changes falls back to the stock implementation and keeps matching
    for m, opt in ((a, ref), (b, mine)):
        opt.zero_grad(set_to_none=True)
        m[3](m[2](m[1](m[0](x)))).square().mean().backward()
        opt.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
