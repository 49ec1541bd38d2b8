"""Fused detection-loss ops over libefg_hip.so (csrc/det_loss.hip): the matching cost and the per-layer focal /
box losses of $CQ/modules/matcher.py:40-80 and $CQ/losses.py:26-108, one kernel per family and direction."""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L


def device_scalar(value, device):
    """A 0-dim float32 tensor on `device` holding `value` (a Python number or already a device tensor)."""
    if torch.is_tensor(value):
        return value.to(device=device, dtype=torch.float32).reshape(())
    return torch.tensor(float(value), dtype=torch.float32).to(device, non_blocking=True)


def match_cost(logits, boxes, tgt_labels, tgt_boxes, w_class, w_bbox, w_giou, w_rad, alpha=0.25, gamma=2.0, out=None):
    """logits [L,B,Q,C], boxes [L,B,Q,7], tgt_labels [B,G] int64, tgt_boxes [B,G,7] -> cost [L*B, Q, G] (into `out`, a
    contiguous [L*B, Q, G] fp32 view, if given)."""
    L.require_gpu(logits, boxes, tgt_labels, tgt_boxes)
    n_layers, b, q, c = logits.shape
    g = tgt_labels.shape[1]
    lg, bx = logits.detach().contiguous().float(), boxes.detach().contiguous().float()
    cost = out if out is not None else torch.empty((n_layers * b, q, g), dtype=torch.float32, device=logits.device)
    assert cost.shape == (n_layers * b, q, g) and cost.dtype == torch.float32 and cost.is_contiguous()
    L.check(L.lib().efg_match_cost_f32(L.ptr(lg), L.ptr(bx), L.ptr(tgt_labels.contiguous()),
                                       L.ptr(tgt_boxes.contiguous().float()), n_layers * b, b, q, c, g, float(w_class),
                                       float(w_bbox), float(w_giou), float(w_rad), float(alpha), float(gamma),
                                       L.ptr(cost), L.stream()))
    return cost


class FocalLossLayers(Function):
    """logits [L, ..., C], target_class int32 [L, ...] (-1 = background) -> [L] sums / denom."""

    @staticmethod
    def forward(ctx, logits, target_class, denom, alpha, gamma):
        n_layers, c = logits.shape[0], logits.shape[-1]
        lg = logits.contiguous()
        n = lg.numel() // (n_layers * c) if n_layers else 0
        out = torch.empty(n_layers, dtype=torch.float32, device=logits.device)
        L.check(L.lib().efg_focal_loss_forward_f32(L.ptr(lg), L.ptr(target_class), n_layers, n, c, float(alpha),
                                                   float(gamma), L.ptr(denom), L.ptr(out), L.stream()))
        ctx.save_for_backward(lg, target_class, denom)
        ctx.cfg = (n_layers, n, c, float(alpha), float(gamma))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        lg, target_class, denom = ctx.saved_tensors
        n_layers, n, c, alpha, gamma = ctx.cfg
        grad = torch.empty_like(lg)
        L.check(L.lib().efg_focal_loss_backward_f32(L.ptr(lg), L.ptr(target_class), n_layers, n, c, alpha, gamma,
                                                    L.ptr(denom), L.ptr(grad_out.contiguous()), L.ptr(grad), L.stream()))
        return grad, None, None, None, None


class BoxLossLayers(Function):
    """boxes [L,B,Q,7], tgt_boxes [B,G,7], pair indices (l, b, q, g) int64 [n] -> [L, 3] (bbox L1, 1 - GIoU, rad L1)
    sums / denom."""

    @staticmethod
    def forward(ctx, boxes, tgt_boxes, l_idx, b_idx, q_idx, g_idx, denom):
        n_layers, b, q = boxes.shape[:3]
        g = tgt_boxes.shape[1]
        bx, tb = boxes.contiguous(), tgt_boxes.contiguous()
        idx = [t.contiguous() for t in (l_idx, b_idx, q_idx, g_idx)]
        n = idx[0].numel()
        out = torch.empty((n_layers, 3), dtype=torch.float32, device=boxes.device)
        L.check(L.lib().efg_box_loss_forward_f32(L.ptr(bx), L.ptr(tb), *[L.ptr(t) for t in idx], n, n_layers, b, q, g,
                                                 L.ptr(denom), L.ptr(out), L.stream()))
        ctx.save_for_backward(bx, tb, *idx, denom)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        bx, tb, l_idx, b_idx, q_idx, g_idx, denom = ctx.saved_tensors
        n_layers, b, q = bx.shape[:3]
        grad = torch.zeros_like(bx)
        L.check(L.lib().efg_box_loss_backward_f32(L.ptr(bx), L.ptr(tb), L.ptr(l_idx), L.ptr(b_idx), L.ptr(q_idx),
                                                  L.ptr(g_idx), l_idx.numel(), n_layers, b, q, tb.shape[1], L.ptr(denom),
                                                  L.ptr(grad_out.contiguous()), L.ptr(grad), L.stream()))
        return grad, None, None, None, None, None, None


class BoxRefineFunction(Function):
    """sigmoid(delta + inverse_sigmoid(anchor)) in one launch each way (csrc/det_loss.hip).  The decoder's reference windows
    are detached between layers ($CQ/transformer.py:331-336) -- no anchor gradient; the model's per-layer heads refine the
    previous layer's UNdetached boxes ($CQ/voxel_detr.py:171-180) -- the same launch also returns the anchor gradient."""

    @staticmethod
    def forward(ctx, delta, anchor, eps):
        d, a = delta.contiguous(), anchor.contiguous()
        out = torch.empty_like(d)
        L.check(L.lib().efg_box_refine_forward_f32(L.ptr(d), L.ptr(a), d.numel(), float(eps), L.ptr(out), L.stream()))
        ctx.eps = float(eps)
        ctx.save_for_backward(out, a if anchor.requires_grad else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        out, anchor = ctx.saved_tensors
        g = grad.contiguous()
        gd = torch.empty_like(out)
        ga = torch.empty_like(out) if anchor is not None and ctx.needs_input_grad[1] else None
        L.check(L.lib().efg_box_refine_backward_f32(L.ptr(g), L.ptr(out), L.ptr(anchor) if ga is not None else None,
                                                    out.numel(), ctx.eps, L.ptr(gd), L.ptr(ga) if ga is not None else None,
                                                    L.stream()))
        return gd, ga, None


def box_refine(delta, anchor, eps=1e-5):
    """(delta + inverse_sigmoid(anchor)).sigmoid() -- $CQ/heads.py:78.  GPU fp32 tensors of equal shape take the fused
    kernel; anything else the PyTorch formulation (same math)."""
    if (delta.is_cuda and delta.dtype == torch.float32 and anchor.dtype == torch.float32 and delta.shape == anchor.shape
            and os.environ.get("EFG_FUSED_LOSS", "1") != "0"):
        return BoxRefineFunction.apply(delta, anchor, eps)
    x = anchor.clamp(min=0, max=1)
    return (delta + torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))).sigmoid()


def topk_unsorted(x, k):
    """(values, indices) of the k largest entries of every row of x [rows, n] (GPU fp32), ascending index order; ties at the
    cut go to the lowest indices.  == torch.topk(x, k, dim=1, sorted=False) as a set."""
    L.require_gpu(x)
    x = x.contiguous()
    rows, n = x.shape
    values = torch.empty((rows, k), dtype=torch.float32, device=x.device)
    indices = torch.empty((rows, k), dtype=torch.int64, device=x.device)
    L.check(L.lib().efg_topk_unsorted_f32(L.ptr(x), rows, n, k, L.ptr(values), L.ptr(indices), L.stream()))
    return values, indices
