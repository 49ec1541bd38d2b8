"""AdamW with the reference's parameter groups ($CQ/modules/optimizer.py:11-70): backbone at
lr_backbone, `linear_box*` (deformable offsets) at lr * deform_lr_multi, everything else at lr."""
import torch


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
        oc.setdefault("fused", True)  # one multi-tensor kernel per group instead of ~40 foreach launches; same update
    return torch.optim.AdamW(groups, **oc)
