"""AdamW with the reference's parameter groups ($CQ/modules/optimizer.py:11-70): backbone at
lr_backbone, `linear_box*` (deformable offsets) at lr * deform_lr_multi, everything else at lr."""
import torch


class CachedFusedAdamW(torch.optim.AdamW):
    """torch.optim.AdamW(fused=True), same state and same `torch._fused_adamw_` update, with the per-step Python
    bookkeeping hoisted out of the loop.  The stock `step()` walks every parameter to rebuild six tensor lists, groups
    them by device / dtype and only then launches the three multi-tensor kernels (~1 ms of host time per step on this
    model's ~300 parameters; the step's host side is within 10 % of its device side).  Here the lists are built once
    (by torch's own `_init_group`, so the state dict is interchangeable) and a step is: collect the current gradients,
    bump the step counters, one fused call per group.  Anything outside that shape (closure, amsgrad, maximize,
    capturable, a parameter that suddenly has no gradient) goes to the stock implementation."""

    def __init__(self, params, **kw):
        kw["fused"] = True
        super().__init__(params, **kw)
        self._plan = None

    def _plain(self, group):
        return not (group["amsgrad"] or group["maximize"] or group["capturable"] or group["differentiable"]
                    or torch.is_tensor(group["lr"]))

    def _build_plan(self):
        plan = []
        for group in self.param_groups:
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            assert len({(p.device, p.dtype) for p in params}) <= 1, "one device / dtype per group"
            plan.append((params, exp_avgs, exp_avg_sqs, steps))
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not all(self._plain(g) for g in self.param_groups):
            return super().step(closure)
        if self._plan is None:
            self._plan = self._build_plan()
        todo = []
        for group, (params, exp_avgs, exp_avg_sqs, steps) in zip(self.param_groups, self._plan):
            grads = [p.grad for p in params]
            if any(g is None for g in grads) or len(params) != sum(p.grad is not None for p in group["params"]):
                self._plan = None  # the set of trained parameters changed: let torch sort it out
                return super().step()
            todo.append((group, params, grads, exp_avgs, exp_avg_sqs, steps))
        # `found_inf` (set by the Trainer: a 1-element device tensor, 1.0 when this step's loss is not finite): the
        # fused kernel then leaves parameters and moments untouched, and the step counters are taken back -- the same
        # device-side guard GradScaler uses; no read-back
        found_inf = getattr(self, "found_inf", None)
        for group, params, grads, exp_avgs, exp_avg_sqs, steps in todo:
            if not params:
                continue
            beta1, beta2 = group["betas"]
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=group["lr"],
                                beta1=beta1, beta2=beta2, weight_decay=group["weight_decay"], eps=group["eps"],
                                maximize=False, grad_scale=None, found_inf=found_inf)
            if found_inf is not None:
                torch._foreach_sub_(steps, [found_inf.reshape(()).to(steps[0].dtype)] * len(steps))
        return None


def build_adamw_multi(cfg, model):
    oc = dict(cfg.solver.optimizer)
    oc.pop("type", None)
    lr = oc["lr"]
    lr_backbone = oc.pop("lr_backbone", lr)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    backbone = [p for p in model.backbone.parameters() if p.requires_grad]
    rest = [p for n, p in named if "backbone" not in n and "linear_box" not in n]
    deform = [p for n, p in named if "linear_box" in n and "backbone" not in n]
    groups = [{"params": backbone, "lr": lr_backbone}, {"params": rest},
              {"params": deform, "lr": lr * cfg.solver.deform_lr_multi}]
    oc["betas"] = tuple(oc["betas"])
    if all(p.is_cuda for g in groups for p in g["params"]):
        # one multi-tensor kernel per group instead of ~40 foreach launches; same update
        return CachedFusedAdamW(groups, **oc)
    return torch.optim.AdamW(groups, **oc)
