"""Golden vectors from the reference's OWN `VoxelDETR.forward` (training mode, 32 loss terms, backward), run on CPU
in the build container at a reduced grid -- SURVEY.md §4 item 5 / Appendix A.5.

What runs is the reference's code, imported in place from /root/reference behind import shims:
`$CQ/voxel_detr.py:VoxelDETR` with its `Backbone3d`, `SparseResNet` wiring (sparse_net.py:284-309),
`FPN.forward` (fpn.py:136-169), `input_proj`, `PositionEmbeddingSine`, `Transformer`, `Det3DHead`, CDN, matcher,
losses and the contrastive double loop (voxel_detr.py:223-254); points are voxelized by the reference's numba
voxelizer (`efg/geometry/point_cloud_ops.py`, identity-jit) and collated by the reference's `collate`.  Two things
are stand-ins, as in the survey: `BoxAttnFunction` -> the reference's `ms_deform_attn_core_pytorch`, and
`spconv.pytorch` -> a dense-masked `F.conv3d` stand-in (the third-party package is absent).  The stand-in pins
everything except spconv's own arithmetic, which tests/test_spconv_dense_gpu.py pins against conv3d directly.

Nothing of the reference is copied: only tensors (inputs, intermediate maps, losses, a few gradients) are saved to
tests/golden/conquer_full_small.npz.  Weights are NOT stored: both sides fill the state dict with
tests/golden_init.py:deterministic_state (same names and shapes => same values).
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"
CQ = REF + "/playground/detection.3d/waymo/conquer/ConQueR.waymo.res18.p3.dn3.tau07.noised_only.bs6.epoch6"
VD = REF + "/playground/detection.3d/waymo/conquer/VoxelDETR.waymo.res18.p3.box_only_with_3cat.bs6.epoch6"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, **attrs):
    m = _mod(name, **attrs)
    m.__path__ = []
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


# ---- dense-masked stand-in for spconv.pytorch (ours) ---------------------------------
class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features, self.indices = features, indices
        self.spatial_shape, self.batch_size = list(spatial_shape), batch_size

    def replace_feature(self, f):
        return SparseConvTensor(f, self.indices, self.spatial_shape, self.batch_size)

    def dense(self):
        d, h, w = self.spatial_shape
        c = self.features.shape[1]
        out = self.features.new_zeros(self.batch_size, d, h, w, c)
        i = self.indices.long()
        out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous()


class SparseModule(nn.Module):
    pass


class SparseSequential(SparseModule):
    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                x = x.replace_feature(m(x.features))
            else:
                x = m(x)
        return x


def _t3(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)


class _Conv(SparseModule):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True, indice_key=None, subm=False):
        super().__init__()
        self.k, self.s, self.p, self.subm = _t3(kernel_size), _t3(stride), _t3(padding), subm
        self.weight = nn.Parameter(torch.randn(cout, *self.k, cin) * 0.05)  # spconv 2.x layout [Cout,kd,kh,kw,Cin]
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def forward(self, x):
        w = self.weight.permute(0, 4, 1, 2, 3)
        dense = x.dense()
        if self.subm:
            y = F.conv3d(dense, w, self.bias, 1, tuple(k // 2 for k in self.k))
            i = x.indices.long()
            return x.replace_feature(y[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]])
        y = F.conv3d(dense, w, self.bias, self.s, self.p)
        occ = torch.zeros_like(dense[:, :1])
        i = x.indices.long()
        occ[i[:, 0], 0, i[:, 1], i[:, 2], i[:, 3]] = 1
        act = F.conv3d(occ, torch.ones(1, 1, *self.k, dtype=occ.dtype), None, self.s, self.p) > 0
        idx = torch.nonzero(act[:, 0])  # sorted (b, z, y, x)
        feats = y[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
        return SparseConvTensor(feats, idx.int(), list(y.shape[2:]), x.batch_size)


class SubMConv3d(_Conv):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        super().__init__(cin, cout, kernel_size, 1, padding, bias, indice_key, subm=True)


class SparseConv3d(_Conv):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        super().__init__(cin, cout, kernel_size, stride, padding, bias, indice_key, subm=False)


def install_shims(model_dir=CQ):
    sys.path.insert(0, REF)
    import efg  # the real (trivial) package root

    c_stub = _mod("efg._C")
    def _c_getattr(name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _missing(*a, **k):
            raise NotImplementedError("efg._C.%s (stub)" % name)

        return _missing

    c_stub.__getattr__ = _c_getattr
    efg._C = c_stub
    _mod("torch._six", string_classes=(str, bytes))
    _pkg("torchvision")
    _mod("portalocker")
    _mod("termcolor", colored=lambda s, *a, **k: s)
    _mod("easydict", EasyDict=dict)
    _pkg("tensorboard")
    _mod("cv2")
    om = _mod("omegaconf")

    class OmegaConf:
        @staticmethod
        def to_container(x, **k):
            return x

    om.OmegaConf, om.DictConfig, om.ListConfig = OmegaConf, dict, list
    # numba (identity jit) for the reference's CPU voxelizer
    nb = _mod("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.jit = nb.njit = jit
    # efg.data drags in pycocotools / cv2 / PIL: provide only what the model imports
    _pkg("efg.data")
    _pkg("efg.data.datasets")
    src = open(REF + "/efg/data/datasets/waymo/waymo.py").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "collate"][0]
    ns = {"collections": __import__("collections"), "np": np, "torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "waymo.collate", "exec"), ns)
    _mod("efg.data.datasets.waymo", collate=ns["collate"])
    _pkg("efg.data.structures")
    _load("efg.data.structures.shape_spec", REF + "/efg/data/structures/shape_spec.py")
    # spconv stand-in
    sp = _pkg("spconv")
    spp = _mod("spconv.pytorch", SparseConvTensor=SparseConvTensor, SparseModule=SparseModule,
               SparseSequential=SparseSequential, SubMConv3d=SubMConv3d, SparseConv3d=SparseConv3d)
    sp.pytorch = spp
    # the sampling op: the reference's own pure-PyTorch core (efg/operators/ms_deform_attn.py:55-76)
    from efg.operators.ms_deform_attn import ms_deform_attn_core_pytorch as core

    class BoxAttnFunction:
        kink = []  # distance of every sampling coordinate (pixel units) to the nearest bilinear kink (an integer)

        @staticmethod
        def apply(value, shapes, start, loc, attn, step):
            b, lq, h, l = attn.shape[:4]
            if loc.requires_grad or value.requires_grad:
                with torch.no_grad():
                    wh = torch.as_tensor(shapes.tolist(), dtype=loc.dtype).flip(-1)  # (W, H) per level
                    pix = loc.detach() * wh[None, None, None, :, None, :] - 0.5
                    d = (pix - pix.round()).abs()
                    inside = (pix > -1) & (pix < wh[None, None, None, :, None, :])
                    BoxAttnFunction.kink.append(torch.where(inside, d, torch.ones_like(d)).flatten().double())
            return core(value, shapes.tolist(), loc, attn.reshape(b, lq, h, l, -1))

    _mod("efg.modeling.operators", BoxAttnFunction=BoxAttnFunction)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return _orig_to(self, *a, **k)

    torch.Tensor.to = _to
    sys.path.insert(0, model_dir)


def make_config(yaml_name, extra=None):
    """Our YAML (same keys as $CQ / $VD config.yaml:67-144) at the reduced grid of Appendix A.5."""
    from golden_init import FULL_OVERRIDES

    from efg_amd.config import load_config

    ov = dict(FULL_OVERRIDES)
    ov["model.device"] = "cpu"
    ov.update(extra or {})
    return load_config(os.path.join(ROOT, "configs", yaml_name), ov)


def reference_samples(cfg, points_list, annos):
    """points -> the reference's Voxelization processor dict (extend_3d.py:267-283) via its numba voxelizer."""
    pco = _load("ref_point_cloud_ops", REF + "/efg/geometry/point_cloud_ops.py")
    vs = np.array(cfg.dataset.voxel_size, np.float32)
    rng = np.array(cfg.dataset.pc_range, np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
    out = []
    for pts, ann in zip(points_list, annos):
        voxels, coords, npv = pco.points_to_voxel(pts, vs, rng, 5, True, 120000)
        sample = {"voxels": voxels, "points": pts, "coordinates": coords, "num_points_per_voxel": npv,
                  "num_voxels": np.array([voxels.shape[0]], np.int64), "shape": grid, "range": rng, "size": vs}
        out.append((sample, {"annotations": {k: np.array(v) for k, v in ann.items()}}))
    return out


def run(tag, model_dir, yaml_name, out_name):
    import copy

    from golden_init import deterministic_state, full_inputs

    install_shims(model_dir)
    import voxel_detr  # the reference model, imported in place

    cfg = make_config(yaml_name)
    torch.manual_seed(0)
    model = voxel_detr.VoxelDETR(cfg)
    state = deterministic_state(model.state_dict())
    model.load_state_dict(state, strict=True)
    # the momentum decoder starts as a copy of the decoder (voxel_detr.py:86-89); deterministic_state keeps that
    model.train()
    points_list, annos = full_inputs()
    batch = reference_samples(cfg, points_list, copy.deepcopy(annos))
    cap = {}
    ext = model.backbone.extractor
    ext.bottom_up.register_forward_hook(lambda m, i, o: cap.update({"bu_" + k: v.detach().clone() for k, v in o.items()}))
    ext.register_forward_hook(lambda m, i, o: cap.update({"fpn_" + k: v.detach().clone() for k, v in o.items()}))
    model.input_proj[0].register_forward_hook(lambda m, i, o: cap.update(src=o.detach().clone()))
    if hasattr(voxel_detr, "prepare_for_cdn"):  # ConQueR only
        orig_cdn = voxel_detr.prepare_for_cdn

        def spy_cdn(*a, **k):
            r = orig_cdn(*a, **k)
            cap.update(dn_label=r[0].detach().clone(), dn_box=r[1].detach().clone(), dn_mask=r[2].clone(),
                       dn_meta={k: v for k, v in r[3].items()})
            return r

        voxel_detr.prepare_for_cdn = spy_cdn
    tr_mod = model.transformer
    tr_mod.register_forward_hook(lambda m, i, o: cap.update(hs=o[0].detach().clone(), memory=o[3].detach().clone(),
                                                            topk=o[5].detach().clone()))
    torch.manual_seed(1234)  # the CDN noise stream (global CPU generator, cdn.py:40-42,63-66)
    losses = model(batch)
    total = sum(v for v in losses.values() if v.requires_grad)
    total.backward()
    kinks = torch.cat(sys.modules["efg.modeling.operators"].BoxAttnFunction.kink)
    print("   bilinear kink distances (pixels): n=%d, smallest %s" % (
        kinks.numel(), ["%.1e" % float(x) for x in torch.sort(kinks)[0][:6]]))
    grads = {}
    want = ["backbone.extractor.bottom_up.stem.conv1.0.weight", "backbone.extractor.bottom_up.res3.0.shortcut.0.weight",
            "backbone.extractor.fpn_lateral3.weight", "backbone.extractor.fpn_output3.weight", "input_proj.0.0.weight",
            "transformer.encoder.layers.0.self_attn.linear_box_weight", "projector.0.weight", "predictor.2.weight",
            "transformer.decoder.layers.1.multihead_attn.value_proj.weight",
            "backbone.extractor.bottom_up.res4.1.conv2.0.weight", "backbone.extractor.bottom_up.res3_out.0.weight",
            # small tensors from the loss inwards (they localise a backward discrepancy)
            "transformer.decoder.detection_head.class_embed.2.layers.2.weight",
            "transformer.decoder.detection_head.bbox_embed.2.layers.2.weight",
            "transformer.decoder.detection_head.bbox_embed.0.layers.0.bias",
            "transformer.decoder.layers.2.norm3.weight", "transformer.decoder.layers.2.linear2.bias",
            "transformer.decoder.layers.2.multihead_attn.linear_box_weight",
            "transformer.decoder.layers.2.multihead_attn.linear_attn_weight",
            "transformer.decoder.layers.2.multihead_attn.out_proj.bias",
            "transformer.decoder.layers.2.multihead_attn.value_proj.bias",
            "transformer.decoder.layers.2.self_attn.out_proj.bias", "transformer.decoder.layers.2.self_attn.in_proj_bias",
            "transformer.decoder.layers.2.norm1.weight", "transformer.decoder.layers.2.norm2.weight",
            "transformer.decoder.layers.0.pos_embed_layer.layers.0.weight",
            "transformer.proposal_head.class_embed.0.layers.2.weight", "transformer.proposal_head.bbox_embed.0.layers.2.weight",
            "transformer.encoder.layers.0.norm2.weight", "transformer.encoder.layers.0.linear2.bias",
            "transformer.encoder.layers.0.self_attn.value_proj.bias", "input_proj.0.1.weight",
            "backbone.extractor.fpn_output3.norm.weight", "backbone.extractor.bottom_up.res4.1.conv.4.weight"]
    params = dict(model.named_parameters())
    for n in want:
        if n in params and params[n].grad is not None:
            grads[n] = params[n].grad.detach().clone()
    dead = sorted(n for n, p in params.items() if p.requires_grad and p.grad is None)
    # ---- the same reference model in float64 (ground truth for the gradient tolerances): same weights, same
    # inputs, the CDN queries of the fp32 run replayed (rand_like on double tensors would draw a different stream)
    grads64 = {}
    if "--no-fp64" not in sys.argv:
        model64 = voxel_detr.VoxelDETR(cfg)
        model64.load_state_dict(state, strict=True)
        model64.double().train()
        model64.box_coder.pc_range = model64.box_coder.pc_range.double() if torch.is_tensor(model64.box_coder.pc_range) else model64.box_coder.pc_range
        batch64 = reference_samples(cfg, points_list, copy.deepcopy(annos))
        for smp, tgt in batch64:
            smp["voxels"] = smp["voxels"].astype(np.float64)
            tgt["annotations"]["gt_boxes"] = tgt["annotations"]["gt_boxes"].astype(np.float64)
        if "dn_box" in cap:
            fixed = (cap["dn_label"].double(), cap["dn_box"].double())

            def replay_cdn(*a, **k):
                return fixed[0], fixed[1], cap["dn_mask"], dict(cap["dn_meta"])

            voxel_detr.prepare_for_cdn = replay_cdn
        torch.manual_seed(1234)
        losses64 = model64(batch64)
        total64 = sum(v for v in losses64.values() if v.requires_grad)
        total64.backward()
        p64 = dict(model64.named_parameters())
        for n in grads:
            grads64[n] = p64[n].grad.detach().clone()
        print("   fp64 total loss %.9f (fp32 %.9f)" % (float(total64), float(total)))
    if "--grad64-out" in sys.argv:
        # the fp64 run's gradients of the same tensors, sliced like the fp32 ones: the ground truth of the measured gradient
        # table (scripts/grad_tier_report.py -> profiles/r06_grad_tier_table.txt); the main fixture is left untouched
        out64 = os.path.join(ROOT, "tests", "golden", sys.argv[sys.argv.index("--grad64-out") + 1])
        save64 = {}
        for k, v in grads64.items():
            save64["grad64::" + k] = (v[:8].contiguous() if v.numel() > 65536 else v).numpy()
            g32 = grads[k]
            save64["ref32_err::" + k] = np.array(float((g32.double() - v).abs().max() / v.abs().max()))
        save64["total_loss64"] = np.array(float(total64))
        np.savez_compressed(out64, **save64)
        print(tag, "saved", out64, os.path.getsize(out64) // 1024, "KiB;", len(grads64), "fp64 gradients")
        return
    save = {"total_loss": total.detach(), "n_params": np.array(sum(p.numel() for p in params.values() if p.requires_grad))}
    light = "prepare_for_cdn" not in vars(voxel_detr)  # Voxel-DETR: same weights => same maps as the ConQueR fixture
    for k in ("memory", "topk") if light else ("bu_res3", "bu_res4", "fpn_p3", "src", "memory", "topk", "dn_label",
                                                "dn_box"):
        if k in cap:
            save[k] = cap[k]
    if not light:
        save["bu_res2_sub"] = cap["bu_res2"][:, ::8].contiguous()  # every 8th channel (3 MB -> 0.4 MB)
        save["fpn_p2_sub"] = cap["fpn_p2"][:, ::16].contiguous()
    for k, v in losses.items():
        save["loss::" + k] = v.detach()
    for k, v in grads.items():
        # large tensors: the first 8 output rows only (the test slices its gradient the same way)
        save["grad::" + k] = v[:8].contiguous() if v.numel() > 65536 else v
        if k in grads64:
            g64 = grads64[k]
            e = float((v.double() - g64).abs().max() / g64.abs().max())
        else:
            e = float("nan")
        print("   grad %-75s %-22s max %.3e   fp32-vs-fp64 err/max %.2e" % (k, tuple(v.shape), float(v.abs().max()), e))
    print("   missing:", [n for n in want if n not in grads])
    bn = model.backbone.extractor.bottom_up.stem.conv1[1]
    save["bn_running_mean_after"] = bn.running_mean.detach().clone()
    save["dead_params"] = np.array(";".join(dead))
    save["min_kink_distance_px"] = np.array(float(kinks.min()))
    out = os.path.join(ROOT, "tests", "golden", out_name)
    np.savez_compressed(out, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in save.items()})
    print(tag, "saved", out, os.path.getsize(out) // 1024, "KiB;", len(losses), "loss terms; total", float(total),
          "; params", int(save["n_params"]), "; dead", len(dead))
    for k, v in sorted(losses.items()):
        print("   %-28s %.6f" % (k, float(v)))


def run_infer(tag, model_dir, yaml_name, out_name):
    """The reference model's INFERENCE branch (eval mode: no CDN, decode, score selection -- ConQueR: every (query,
    class) with score >= 0.1, one scene per call; Voxel-DETR: the 300 best pairs) on the same weights and scenes."""
    from golden_init import deterministic_state, full_inputs

    install_shims(model_dir)
    import voxel_detr  # the reference model, imported in place

    from golden_init import INFER_OVERRIDES

    cfg = make_config(yaml_name, INFER_OVERRIDES)
    torch.manual_seed(0)
    model = voxel_detr.VoxelDETR(cfg)
    model.load_state_dict(deterministic_state(model.state_dict()), strict=True)
    model.eval()
    points_list, annos = full_inputs()
    save = {}
    with torch.no_grad():
        for i in range(len(points_list)):
            res = model(reference_samples(cfg, points_list[i:i + 1], annos[i:i + 1]))[0]
            for k in ("scores", "labels", "boxes3d"):
                save["%s::%d" % (k, i)] = res[k].numpy()
            print("   %s scene %d: %d detections, scores %.3f..%.3f" % (tag, i, len(res["scores"]), float(res["scores"].min()),
                                                                float(res["scores"].max())))
    out = os.path.join(ROOT, "tests", "golden", out_name)
    np.savez_compressed(out, **save)
    print(tag, "saved", out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "conquer"
    if "--infer" in sys.argv:
        if which == "conquer":
            run_infer("ConQueR", CQ, "conquer_waymo_res18.yaml", "conquer_infer_small.npz")
        else:
            run_infer("VoxelDETR", VD, "voxeldetr_waymo_res18.yaml", "voxeldetr_infer_small.npz")
    elif which == "conquer":
        run("ConQueR", CQ, "conquer_waymo_res18.yaml", "conquer_full_small.npz")
    else:
        run("VoxelDETR", VD, "voxeldetr_waymo_res18.yaml", "voxeldetr_full_small.npz")
