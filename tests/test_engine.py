"""Trainer step = the reference's step (efg/engine/trainer.py:278-305, hooks.py:68-81,118-121): OneCycle learning
rate + cycled Adam beta1 as efg/solver/lr_schedulers.py:222-237 builds them (max_lr = solver.optimizer.lr for every
parameter group), optional gradient clipping, FloatingPointError on a non-finite loss.  CPU, oracle ops."""
import math

import pytest
import torch


def _trainer(**kw):
    from efg_amd.engine import Trainer

    ov = {"model.transformer.num_queries": 40, "model.transformer.enc_layers": 1,
          "dataset.pc_range": [-12.8, -12.8, -2.0, 12.8, 12.8, 4.0]}
    ov.update(kw.pop("overrides", {}))
    return Trainer(device="cpu", overrides=ov, seed=0, ddp=False, **kw)


def _batch():
    from test_model_parity_gpu import _scene

    out = []
    for i in range(1):
        pts, ann = _scene(900 + i, n=3000)
        out.append(({"points": torch.from_numpy(pts)}, {"annotations": ann}))
    return out


def test_one_cycle_schedule_matches_torch_onecycle(oracle_mod):
    from oracle import cpu_backend

    tr = _trainer(max_iters=10)
    assert tr.lr_scheduler is not None
    ref_opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3, betas=(0.9, 0.99))
    ref = torch.optim.lr_scheduler.OneCycleLR(ref_opt, 1e-3, total_steps=10, pct_start=0.4, base_momentum=0.85,
                                              max_momentum=0.95, div_factor=10.0)
    # every group starts at max_lr / div_factor -- the scalar max_lr overrides AdamWMulti's per-group rates
    assert all(math.isclose(g["lr"], 1e-4, rel_tol=1e-9) for g in tr.optimizer.param_groups)
    with cpu_backend.install():
        for _ in range(3):
            tr.step(_batch())
            ref_opt.step()
            ref.step()
            for g in tr.optimizer.param_groups:
                assert math.isclose(g["lr"], ref_opt.param_groups[0]["lr"], rel_tol=1e-9)
                assert math.isclose(g["betas"][0], ref_opt.param_groups[0]["betas"][0], rel_tol=1e-9)
    tr.close()


def test_non_finite_loss_raises_like_the_reference(oracle_mod):
    from oracle import cpu_backend

    tr = _trainer(max_iters=10)
    with torch.no_grad():  # poisons the contrastive terms only: matching stays feasible, the summed loss is NaN
        tr.model.projector[0].weight.fill_(float("nan"))
    with cpu_backend.install(), pytest.raises(FloatingPointError):
        tr.step(_batch())  # the first step is always checked


@pytest.mark.gpu
def test_infeasible_matching_is_loud_and_safe_gpu(dev):
    """NaN class logits make every matching cost non-finite: scipy raises in the reference (matcher.py:89).  On the
    device the assignment comes back unmatched (-1); the loss kernels must not index with it (out-of-bounds
    write in box_loss_grad_kernel) and the engine must raise at its next check."""
    from efg_amd.engine import Trainer

    ov = {"model.transformer.num_queries": 40, "model.transformer.enc_layers": 1,
          "dataset.pc_range": [-12.8, -12.8, -2.0, 12.8, 12.8, 4.0]}
    tr = Trainer(device=dev, overrides=ov, seed=0, ddp=False, max_iters=10)
    with torch.no_grad():
        for m in tr.model.transformer.decoder.detection_head.class_embed:
            m.layers[0].weight.fill_(float("nan"))
    batch = [({"points": b[0]["points"].to(dev)}, b[1]) for b in _batch()]
    with pytest.raises(FloatingPointError):
        tr.step(batch)
    torch.cuda.synchronize()  # no fault pending from an out-of-range access
    tr.close() if tr._nonfinite is None else None


def test_grad_clipper_norm(oracle_mod):
    from oracle import cpu_backend

    tr = _trainer(max_iters=10, overrides={"solver.grad_clipper.enabled": True,
                                           "solver.grad_clipper.params.max_norm": 0.5})
    seen = {}
    orig = tr.optimizer.step

    def spy(*a, **k):
        seen["norm"] = float(torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in tr.model.parameters()
                                                                   if p.grad is not None])))
        return orig(*a, **k)

    tr.optimizer.step = spy
    with cpu_backend.install():
        tr.step(_batch())
    assert seen["norm"] <= 0.5 * (1 + 1e-4)


@pytest.mark.gpu
def test_non_finite_step_never_reaches_the_weights_gpu(dev):
    """On the GPU the non-finite-loss error is reported at the next periodic check (no per-step read-back), but the
    update itself is skipped on the device: parameters, Adam moments and step counters after a NaN step are those of
    before it."""
    from efg_amd.engine import Trainer

    ov = {"model.transformer.num_queries": 40, "model.transformer.enc_layers": 1,
          "dataset.pc_range": [-12.8, -12.8, -2.0, 12.8, 12.8, 4.0]}
    tr = Trainer(device=dev, overrides=ov, seed=0, ddp=False, max_iters=10)
    batch = [({"points": b[0]["points"].to(dev)}, b[1]) for b in _batch()]
    tr.step(batch)                                   # a clean step (also the always-checked first one)
    before = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
    state = tr.optimizer.state[next(iter(tr.model.transformer.decoder.parameters()))]
    m_before, s_before = state["exp_avg"].clone(), state["step"].clone()
    with torch.no_grad():
        saved = tr.model.projector[0].weight.clone()
        tr.model.projector[0].weight.fill_(float("nan"))   # contrastive terms NaN: matching stays feasible, loss is NaN
    tr.step(batch)                                   # step 2: not a checked step -> no exception yet
    torch.cuda.synchronize()
    for n, p in tr.model.named_parameters():
        if n == "projector.0.weight" or ".decoder_gt." in n:   # (the momentum decoder is an EMA updated in forward)
            continue
        assert torch.equal(p.detach(), before[n]), n
    assert torch.equal(state["exp_avg"], m_before) and torch.equal(state["step"], s_before)
    with torch.no_grad():
        tr.model.projector[0].weight.copy_(saved)
    with pytest.raises(FloatingPointError):          # ... and the failure is still reported
        tr.close()
