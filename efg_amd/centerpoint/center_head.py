"""CenterPoint detection head, losses and decoding ($CP1/center_head.py:18-386, $CP1/centernet_loss.py:8-56).

Module names follow the reference (`shared_conv.{0,1}`, `tasks.T.<head>.{0,1,3}`) so checkpoints load by name.  The
losses are restated without host round trips: the reference tests `num_pos == 0` on the host every step
(centernet_loss.py:54); here that is a `torch.where` on the device."""
import copy
import math
import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ..modeling.common import get_norm
from ..operators.batchnorm import BatchNormActFunction, run_sequential
from ..operators.iou3d_nms import nms_gpu


def _gather_at(feat, ind):
    """feat [B, C, H, W], ind [B, M] flat (y * W + x) -> [B, M, C]: the predictions at the object centres
    (center_utils.py:_transpose_and_gather_feat)."""
    b, c = feat.shape[:2]
    flat = feat.reshape(b, c, -1)
    return flat.gather(2, ind.unsqueeze(1).expand(-1, c, -1)).transpose(1, 2)


class RegLoss(nn.Module):
    """L1 between the predicted and target box codes at the object centres, per code dimension, normalised by the
    number of objects of the batch (centernet_loss.py:8-28).  Returns [dim]."""

    def forward(self, output, mask, ind, target):
        pred = _gather_at(output, ind)
        m = mask.to(pred.dtype).unsqueeze(2)
        per = (pred * m - target * m).abs()
        return per.sum(dim=(0, 1)) / (m.sum() + 1e-4)


class FastFocalLoss(nn.Module):
    """CornerNet focal loss on a sigmoid heat map: negatives over the whole map weighted (1 - target)^4, positives
    only at the object centres (centernet_loss.py:31-56)."""

    def forward(self, out, target, ind, mask, cat):
        mask = mask.to(out.dtype)
        neg = (torch.log(1 - out) * out.pow(2) * (1 - target).pow(4)).sum()
        pos_pred = _gather_at(out, ind).gather(2, cat.unsqueeze(2))  # [B, M, 1]
        pos = (torch.log(pos_pred) * (1 - pos_pred).pow(2) * mask.unsqueeze(2)).sum()
        num_pos = mask.sum()
        # no positives: -neg (the reference branches on the host); else -(pos + neg) / num_pos
        return torch.where(num_pos > 0, -(pos + neg) / num_pos.clamp(min=1), -neg)


# The task stacks of a SepHead as one 320-wide stack (SepHead._forward_fused): A/B switch of the head, default on.
_HEAD_FUSE = os.environ.get("EFG_HEAD_FUSE", "1") != "0"


class _BlockDiagonalWeight(Function):
    """[k_i, C, kh, kw] x n  ->  [sum k_i, n * C, kh, kw] with weight i in rows (sum k_<i ..) and input channels
    (i * C ..), zeros elsewhere; backward hands every weight its block of the joint gradient (one multi-tensor copy
    each way)."""

    @staticmethod
    def forward(ctx, *ws):
        c = ws[0].shape[1]
        ks = [w.shape[0] for w in ws]
        joint = ws[0].new_zeros((sum(ks), c * len(ws)) + tuple(ws[0].shape[2:]))
        if ws[0].dim() == 4 and ws[0].is_contiguous(memory_format=torch.channels_last) and not ws[0].is_contiguous():
            joint = joint.contiguous(memory_format=torch.channels_last)
        torch._foreach_copy_(_BlockDiagonalWeight._blocks(joint, ks, c), [w.detach() for w in ws])
        ctx.ks, ctx.c = ks, c
        return joint

    @staticmethod
    def _blocks(joint, ks, c):
        out, o = [], 0
        for i, k in enumerate(ks):
            out.append(joint[o:o + k, i * c:(i + 1) * c])
            o += k
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        blocks = _BlockDiagonalWeight._blocks(g, ctx.ks, ctx.c)
        outs = [torch.empty_like(b, memory_format=torch.preserve_format).contiguous(
            memory_format=torch.channels_last if g.is_contiguous(memory_format=torch.channels_last) and not g.is_contiguous()
            else torch.contiguous_format) for b in blocks]
        torch._foreach_copy_(outs, blocks)
        return tuple(outs)


class SepHead(nn.Module):
    """One small conv stack per regression target + the heat map ($CP1/center_head.py:18-52)."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, bn=None, init_bias=-2.19):
        super().__init__()
        self.heads = heads
        pad = final_kernel // 2
        for name, (channels, num_conv) in heads.items():
            layers = []
            for _ in range(num_conv - 1):
                layers.append(nn.Conv2d(in_channels, head_conv, final_kernel, stride=1, padding=pad, bias=True))
                if bn is not None:
                    layers.append(get_norm(bn, head_conv))
                layers.append(nn.ReLU())
            layers.append(nn.Conv2d(head_conv, channels, final_kernel, stride=1, padding=pad, bias=True))
            stack = nn.Sequential(*layers)
            if "hm" in name:
                stack[-1].bias.data.fill_(init_bias)  # sigmoid(-2.19) ~ 0.1: a quiet heat map at the start
            else:
                for m in stack.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                        nn.init.constant_(m.bias, 0)
            setattr(self, name, stack)

    def forward(self, x):
        if _HEAD_FUSE and self._fusable(x):
            return self._forward_fused(x)
        return {name: run_sequential(getattr(self, name), x) for name in self.heads}

    # ---- the stacks as ONE stack ----------------------------------------------------------------------------------
    # Every stack reads the same map: conv3x3(C -> 64) + BN + ReLU + conv3x3(64 -> k), k = 2 / 1 / 3 / 2 / 3 on Waymo.  Ten
    # convolutions of 64 output channels or fewer, each with its own BatchNorm launches forward and backward, fill a
    # fraction of the chip.  Here: the first convolutions as one of 5 x 64 output channels (weights concatenated along
    # Cout), ONE BatchNorm + ReLU over the 320 channels (batch statistics are per channel: the same numbers), the
    # second ones as one convolution with a block-diagonal weight [sum k, 320, 3, 3] (the products with the zero blocks
    # are exact zeros).  The modules, their parameters, buffers and state-dict names stay the reference's; the
    # concatenations are differentiable, so every parameter receives its own gradient (a slice of the joint one).
    # 2.17 -> 1.77 ms forward + backward on [2, 64, 188, 188] (scripts/ubench/head_fuse.py).
    def _fusable(self, x):
        stacks = [getattr(self, name) for name in self.heads]
        first = stacks[0]
        if len(stacks) < 2 or x.dim() != 4:
            return False
        for s in stacks:
            if len(s) != 4 or type(s[0]) is not nn.Conv2d or type(s[1]) is not nn.BatchNorm2d or type(s[2]) is not nn.ReLU \
                    or type(s[3]) is not nn.Conv2d:
                return False
            c1, bn, c2 = s[0], s[1], s[3]
            if (c1.kernel_size, c1.padding, c1.stride, c1.dilation, c1.groups) != \
                    (first[0].kernel_size, first[0].padding, (1, 1), (1, 1), 1) or c1.bias is None or c2.bias is None:
                return False
            if (c2.kernel_size, c2.padding, c2.stride, c2.dilation, c2.groups) != \
                    (first[3].kernel_size, first[3].padding, (1, 1), (1, 1), 1) or c1.padding_mode != "zeros" \
                    or c2.padding_mode != "zeros":
                return False
            if c1.out_channels != first[0].out_channels or not bn.affine or not bn.track_running_stats \
                    or bn.momentum is None or (bn.eps, bn.momentum, bn.training) != (first[1].eps, first[1].momentum,
                                                                                     first[1].training):
                return False
        return True

    def _forward_fused(self, x):
        names = list(self.heads)
        stacks = [getattr(self, name) for name in names]
        conv1, bns, conv2 = [s[0] for s in stacks], [s[1] for s in stacks], [s[3] for s in stacks]
        mid = conv1[0].out_channels
        y = F.conv2d(x, torch.cat([c.weight for c in conv1]), torch.cat([c.bias for c in conv1]), padding=conv1[0].padding)
        gamma, beta = torch.cat([bn.weight for bn in bns]), torch.cat([bn.bias for bn in bns])
        mean, var = torch.cat([bn.running_mean for bn in bns]), torch.cat([bn.running_var for bn in bns])
        training = bns[0].training
        if (training and y.is_cuda and y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last)
                and y.shape[1] <= 1024 and os.environ.get("EFG_FUSED_BN", "1") != "0"):
            b, c, h, w = y.shape
            rows = BatchNormActFunction.apply(y.permute(0, 2, 3, 1).reshape(b * h * w, c), None, gamma, beta, mean, var, None,
                                              bns[0].momentum, bns[0].eps, True)
            y = rows.view(b, h, w, c).permute(0, 3, 1, 2)
        else:
            y = F.relu(F.batch_norm(y, mean, var, gamma, beta, training, bns[0].momentum, bns[0].eps))
        if training:
            with torch.no_grad():   # the joint running statistics back into the modules' own buffers: one multi-tensor copy
                torch._foreach_copy_([bn.running_mean for bn in bns] + [bn.running_var for bn in bns],
                                     list(mean.split(mid)) + list(var.split(mid)))
                torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
        out = F.conv2d(y, _BlockDiagonalWeight.apply(*[c.weight for c in conv2]), torch.cat([c.bias for c in conv2]),
                       padding=conv2[0].padding)
        res, o = {}, 0
        for name, c in zip(names, conv2):
            res[name] = out[:, o:o + c.out_channels]
            o += c.out_channels
        return res


class CenterHead(nn.Module):
    def __init__(self, config, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2):
        super().__init__()
        head = config.model.head
        self.class_names = [t["class_names"] for t in head.tasks]
        self.num_classes = [len(n) for n in self.class_names]
        self.code_weights = list(head.misc.code_weights)
        self.weight = head.misc.weight  # heat-map loss vs box loss
        self.dataset = head.misc.dataset
        self.common_heads = {k: tuple(v) for k, v in dict(head.misc.common_heads).items()}
        self.in_channels = head.in_channels
        self.box_n_dim = 9 if "vel" in self.common_heads else 7
        norm = config.model.neck.norm
        self.criterion, self.criterion_reg = FastFocalLoss(), RegLoss()
        self.shared_conv = nn.Sequential(nn.Conv2d(self.in_channels, share_conv_channel, 3, padding=1, bias=True),
                                         get_norm(norm, share_conv_channel), nn.ReLU(inplace=True))
        self.tasks = nn.ModuleList()
        for n in self.num_classes:
            heads = copy.deepcopy(self.common_heads)
            heads["hm"] = (n, num_hm_conv)
            self.tasks.append(SepHead(share_conv_channel, heads, bn=norm, init_bias=init_bias, final_kernel=3))

    def forward(self, x):
        x = run_sequential(self.shared_conv, x)
        return [task(x) for task in self.tasks]

    @staticmethod
    def _sigmoid(x):
        return torch.clamp(x.sigmoid_(), min=1e-4, max=1 - 1e-4)

    def loss(self, example, preds_dicts):
        """{"<task>_loss", "<task>_hm_loss", "<task>_loc_loss", "<task>_num_positive"} ($CP1/center_head.py:115-171)."""
        out = {}
        for t, preds in enumerate(preds_dicts):
            preds["hm"] = self._sigmoid(preds["hm"])
            hm_loss = self.criterion(preds["hm"], example["hm"][t], example["ind"][t], example["mask"][t],
                                     example["cat"][t])
            target_box = example["anno_box"][t]
            parts = [preds["reg"], preds["height"], preds["dim"]]
            if "vel" in preds:
                parts.append(preds["vel"])
            else:
                target_box = torch.cat((target_box[..., :6], target_box[..., -2:]), dim=-1)  # drop the velocity target
            parts.append(preds["rot"])
            preds["anno_box"] = torch.cat(parts, dim=1)
            box_loss = self.criterion_reg(preds["anno_box"], example["mask"][t], example["ind"][t], target_box)
            # the code weights live on the device: `new_tensor(list)` is a pageable upload, i.e. a host wait for everything
            # queued on the stream so far (2.5 ms per step, scripts/ubench/tf_timeline.py --model centerpoint)
            cw = getattr(self, "_code_weights_dev", None)
            if cw is None or cw.device != box_loss.device or cw.dtype != box_loss.dtype:
                cw = self._code_weights_dev = torch.tensor(self.code_weights, dtype=box_loss.dtype, device=box_loss.device)
            loc_loss = (box_loss * cw).sum()
            out["%d_loss" % t] = hm_loss + self.weight * loc_loss
            out["%d_hm_loss" % t] = hm_loss.detach()
            out["%d_loc_loss" % t] = loc_loss
            out["%d_num_positive" % t] = example["mask"][t].float().sum()
        return out

    # ---- inference --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, example, preds_dicts, test_config):
        """Decode every BEV cell into a box, threshold, rotated NMS on the GPU (csrc/iou3d_nms.hip); the plain path of
        $CP1/center_head.py:173-386 (no double-flip test-time augmentation, no circular NMS)."""
        if test_config.get("double_flip", False) or test_config.get("circular_nms", False):
            raise NotImplementedError("double_flip / circular_nms inference is not mirrored")
        dev = preds_dicts[0]["hm"].device
        limit = test_config.post_center_limit_range
        limit = torch.tensor(limit, dtype=torch.float32, device=dev) if len(limit) > 0 else None
        per_task = []
        for preds in preds_dicts:
            p = {k: v.permute(0, 2, 3, 1).contiguous() for k, v in preds.items()}  # N H W C
            hm = torch.sigmoid(p["hm"])
            b, h, w, ncls = hm.shape
            dim = torch.exp(p["dim"]).reshape(b, h * w, 3)
            rot = torch.atan2(p["rot"][..., 0:1], p["rot"][..., 1:2]).reshape(b, h * w, 1)
            reg = p["reg"].reshape(b, h * w, 2)
            hei = p["height"].reshape(b, h * w, 1)
            ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=hm.dtype),
                                    torch.arange(w, device=dev, dtype=hm.dtype), indexing="ij")
            xs = (xs.reshape(1, -1, 1) + reg[..., 0:1]) * test_config.out_size_factor * test_config.voxel_size[0] + \
                test_config.pc_range[0]
            ys = (ys.reshape(1, -1, 1) + reg[..., 1:2]) * test_config.out_size_factor * test_config.voxel_size[1] + \
                test_config.pc_range[1]
            cols = [xs, ys, hei, dim]
            if "vel" in p:
                cols.append(p["vel"].reshape(b, h * w, 2))
            boxes = torch.cat(cols + [rot], dim=2)
            per_task.append(self._post_process(boxes, hm.reshape(b, h * w, ncls), test_config, limit))
        results = []
        for i in range(len(per_task[0])):
            offset, scores, labels, boxes = 0, [], [], []
            for t, n in enumerate(self.num_classes):
                r = per_task[t][i]
                scores.append(r["scores"])
                labels.append(r["label_preds"] + offset)
                boxes.append(r["box3d_lidar"])
                offset += n
            results.append({"scores": torch.cat(scores).cpu(), "labels": (torch.cat(labels) + 1).cpu(),
                            "boxes3d": torch.cat(boxes).cpu()})
        return results

    @staticmethod
    def _post_process(batch_boxes, batch_hm, cfg, limit):
        out = []
        for boxes, hm in zip(batch_boxes, batch_hm):
            scores, labels = hm.max(dim=-1)
            keep = scores > cfg.score_threshold
            if limit is not None:
                keep &= (boxes[:, :3] >= limit[:3]).all(1) & (boxes[:, :3] <= limit[3:]).all(1)
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
            if boxes.shape[0]:
                # the reference hands pcdet-convention boxes to the rotated NMS (box_torch_ops.py:236-258):
                # (x, y, z, w, l, h, -theta - pi/2)
                nms_boxes = torch.stack((boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 4], boxes[:, 3], boxes[:, 5],
                                         -boxes[:, -1] - math.pi / 2), dim=1)
                sel, _ = nms_gpu(nms_boxes.float(), scores.float(), cfg.nms.nms_iou_threshold,
                                 pre_maxsize=cfg.nms.nms_pre_max_size)
                sel = sel[: cfg.nms.nms_post_max_size]
            else:
                sel = torch.zeros(0, dtype=torch.int64, device=boxes.device)
            out.append({"box3d_lidar": boxes[sel], "scores": scores[sel], "label_preds": labels[sel]})
        return out
