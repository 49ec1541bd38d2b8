"""Golden vectors for the CALLER of the hot path: the reference's own ConQueR Transformer / heads /
CDN / losses (playground/detection.3d/waymo/conquer/ConQueR.../{transformer,heads,cdn,losses}.py and
modules/*) run on CPU in the build container, with `BoxAttnFunction` routed to the reference's
ms_deform_attn_core_pytorch.  Stores weights (reference state-dict names), inputs and outputs for a
reduced configuration in tests/golden/conquer_transformer_small.npz.

Nothing of the reference is copied: modules are imported in place behind import shims
 and only tensors are saved.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CQ = REF + "/playground/detection.3d/waymo/conquer/ConQueR.waymo.res18.p3.dn3.tau07.noised_only.bs6.epoch6"
OUT = os.path.join(ROOT, "tests", "golden", "conquer_transformer_small.npz")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install_shims():
    efg = _mod("efg")
    efg.__path__ = []
    _mod("efg._C")
    msda = _load("efg_ref_msda", REF + "/efg/operators/ms_deform_attn.py") if False else None
    # reference pure-PyTorch sampling core (efg/operators/ms_deform_attn.py:55-76), loaded with a stub _C
    efg._C = sys.modules["efg._C"]
    core_mod = _load("efg.operators.ms_deform_attn", REF + "/efg/operators/ms_deform_attn.py")
    core = core_mod.ms_deform_attn_core_pytorch

    class BoxAttnFunction:
        @staticmethod
        def apply(value, shapes, start, loc, attn, step):
            b, lq, h, l = attn.shape[:4]
            return core(value, shapes.tolist(), loc, attn.reshape(b, lq, h, l, -1))

    _mod("efg.modeling").__path__ = []
    _mod("efg.modeling.operators", BoxAttnFunction=BoxAttnFunction)
    _mod("efg.modeling.losses").__path__ = []
    _load("efg.modeling.losses.focal_loss", REF + "/efg/modeling/losses/focal_loss.py")
    _mod("efg.utils").__path__ = []
    _mod("efg.utils.distributed", get_world_size=lambda: 1, reduce_dict=lambda d, average=True: d)
    _mod("torch._six", string_classes=(str, bytes))
    _mod("torchvision")
    torch.Tensor.cuda = lambda self, *a, **k: self
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        return _orig_to(self, *a, **k)

    torch.Tensor.to = _to
    sys.path.insert(0, CQ)


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def cfg_of(d):
    return Cfg({k: cfg_of(v) if isinstance(v, dict) else v for k, v in d.items()})


def main():
    install_shims()
    import cdn
    import heads
    import transformer

    torch.manual_seed(0)
    D, NH, FF, NQ, NC = 64, 4, 128, 20, 3
    config = cfg_of({"model": {"hidden_dim": D, "loss": {"bbox_loss_coef": 4, "giou_loss_coef": 2, "class_loss_coef": 1,
                                                        "rad_loss_coef": 4,
                                                        "matcher": {"class_weight": 1, "bbox_weight": 4, "giou_weight": 2,
                                                                    "rad_weight": 4}},
                               "metrics": [{"type": "accuracy", "params": {}}],
                               "transformer": {"dec_layers": 2}}})
    tr = transformer.Transformer(d_model=D, nhead=NH, nlevel=1, num_encoder_layers=1, num_decoder_layers=2,
                                 dim_feedforward=FF, dropout=0.0, num_queries=NQ, num_classes=NC, mom=0.999)
    tr.proposal_head = heads.Det3DHead(config, with_aux=False, with_metrics=False, num_classes=1, num_layers=1)
    tr.decoder.detection_head = heads.Det3DHead(config, with_aux=True, with_metrics=True, num_classes=NC, num_layers=2)
    import copy
    tr.decoder_gt = copy.deepcopy(tr.decoder)
    for p in tr.decoder_gt.parameters():
        p.requires_grad = False
    # make the zero-initialised attention / box-refinement weights non-trivial so the test has teeth
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in tr.named_parameters():
            if "linear_box_weight" in n or "linear_attn_weight" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            if "bbox_embed" in n and n.endswith("layers.2.weight"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    state = {k: v.detach().clone() for k, v in tr.state_dict().items()}
    B, H, W = 2, 12, 10
    src = torch.randn(B, D, H, W, generator=g)
    pos = torch.randn(B, D, H, W, generator=g) * 0.3
    targets = []
    for n in (3, 2):
        boxes = torch.rand(n, 7, generator=g) * 0.5 + 0.2
        boxes[:, 3:6] = boxes[:, 3:6] * 0.2 + 0.02
        targets.append({"labels": torch.randint(0, NC, (n,), generator=g), "gt_boxes": boxes})
    torch.manual_seed(3)
    lab, box, mask, dn_meta = cdn.prepare_for_cdn((targets, 2, 0.5, 0.4), True, NQ, NC, D, None)
    tr.train()
    hs, init_ref, inter_ref, memory, anchors, topk = tr([src], [pos], box, lab, mask, targets=targets)
    state_after = {k: v.detach().clone() for k, v in tr.state_dict().items() if k.startswith("decoder_gt.")}
    head = tr.decoder.detection_head
    oc, ob = [], []
    for i in range(hs.shape[0]):
        ref = init_ref if i == 0 else inter_ref[i - 1]
        c, b_ = head(hs[i], ref, i)
        oc.append(c)
        ob.append(b_)
    oc, ob = torch.stack(oc), torch.stack(ob)
    dn_meta2 = dict(dn_meta)
    oc2, ob2 = cdn.dn_post_process(oc, ob, dn_meta2, True, lambda a, b_: [{"pred_logits": x, "pred_boxes": y}
                                                                           for x, y in zip(a, b_)])
    outputs = {"pred_logits": oc2[-1][:, :NQ], "pred_boxes": ob2[-1][:, :NQ],
               "aux_outputs": [{"pred_logits": oc2[0][:, :NQ], "pred_boxes": ob2[0][:, :NQ]}]}
    dec_losses = head.compute_losses(outputs, targets, dn_meta2)
    enc_class, enc_coords = tr.proposal_head(memory, anchors)
    bin_targets = copy.deepcopy(targets)
    [t["labels"].fill_(0) for t in bin_targets]
    enc_losses = tr.proposal_head.compute_losses({"topk_indexes": topk, "pred_logits": enc_class,
                                                  "pred_boxes": enc_coords}, bin_targets)
    total = sum(v for v in list(dec_losses.values()) + list(enc_losses.values()) if v.requires_grad)
    total.backward()
    grads = {n: p.grad.detach().clone() for n, p in tr.named_parameters() if p.grad is not None and (
        "layers.0.self_attn.linear_box_weight" in n or "encoder.layers.0.self_attn.value_proj.weight" in n
        or "decoder.layers.1.multihead_attn.linear_attn_weight" in n or "decoder.layers.0.pos_embed_layer.layers.0.weight" in n)}
    save = {"src": src, "pos": pos, "dn_label": lab, "dn_box": box, "dn_mask": mask, "memory": memory, "hs": hs,
            "init_ref": init_ref, "inter_ref": inter_ref, "topk": topk, "logits": oc, "boxes": ob,
            "total_loss": total.detach()}
    for i, t in enumerate(targets):
        save["tgt%d_labels" % i] = t["labels"]
        save["tgt%d_boxes" % i] = t["gt_boxes"]
    for k, v in state.items():
        save["w::" + k] = v
    for k, v in state_after.items():
        save["wgt_after::" + k] = v
    for k, v in {**dec_losses, **{a + "_enc": b_ for a, b_ in enc_losses.items()}}.items():
        save["loss::" + k] = v.detach()
    for k, v in grads.items():
        save["grad::" + k] = v
    np.savez_compressed(OUT, **{k: v.detach().numpy() if torch.is_tensor(v) else np.asarray(v) for k, v in save.items()})
    print("saved", OUT, os.path.getsize(OUT) // 1024, "KiB;", len(dec_losses) + len(enc_losses), "losses; total",
          float(total))


if __name__ == "__main__":
    main()
