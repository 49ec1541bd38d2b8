"""`efg._C` on MI355X: the Python-visible functions of the reference's pybind module
(efg/operators/src/vision.cpp:70-122) for the hot path, with the reference's argument lists, implemented over
libefg_hip.so (C ABI, include/efg_hip.h).  `efg_amd.compat.install()` registers this module as `efg._C`, so reference
wrappers that do `from efg import _C` / `from efg._C import ...` at import time
(efg/operators/{voxelize,scatter_points,box_attention_func,ms_deform_attn,iou3d_nms,deform_conv}.py) import cleanly.

In scope (12 of the 32 bindings): hard_voxelize, dynamic_voxelize, dynamic_point_to_voxel_forward / _backward,
box_attn_forward / _backward, ms_deform_attn_forward / _backward, boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu,
nms_normal_gpu.  Every other name of vision.cpp (COCOeval, deform_conv, knn, swin window ops, ...: 2-D / Mask2Former /
pytorch3d utilities that the ConQueR / CenterPoint train step never reaches) resolves to a
function that raises NotImplementedError WHEN CALLED, so importing modules that merely reference them works.
"""
import torch

from .operators.box_attention_func import (box_attn_backward, box_attn_forward, ms_deform_attn_backward,  # noqa: F401
                                           ms_deform_attn_forward)
from .operators.scatter_points import dynamic_point_to_voxel_backward, dynamic_point_to_voxel_forward  # noqa: F401
from .operators.voxelize import dynamic_voxelize, hard_voxelize  # noqa: F401

_OUT_OF_SCOPE = (
    "COCOevalAccumulate", "COCOevalEvaluateImages", "InstanceAnnotation", "ImageEvaluation", "iou_box3d",
    "sort_vertices_forward", "roll_and_window_partition_forward", "roll_and_window_partition_backward",
    "window_merge_and_roll_forward", "window_merge_and_roll_backward", "boxes_iou_bev_cpu", "box_iou_rotated",
    "deform_conv_forward", "deform_conv_backward_input", "deform_conv_backward_filter",
    "modulated_deform_conv_forward", "modulated_deform_conv_backward", "knn_check_version", "knn_points_idx",
    "knn_points_backward", "nms_rotated")


def get_compiler_version():
    return "hipcc (gfx950)"


def get_cuda_version():
    """vision.cpp:76 reports the CUDA runtime; there is none -- the HIP version PyTorch was built with."""
    return int("".join(ch for ch in (torch.version.hip or "0").split("-")[0] if ch.isdigit())[:5] or 0)


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """(N,7), (M,7), out (N,M): rotated-rectangle intersection area, written into `ans_overlap`
    (iou3d_nms_api / iou3d_nms.py:56)."""
    from .operators.iou3d_nms import boxes_overlap_bev

    ans_overlap.copy_(boxes_overlap_bev(boxes_a, boxes_b))
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    from .operators.iou3d_nms import boxes_iou_bev

    ans_iou.copy_(boxes_iou_bev(boxes_a, boxes_b))
    return 1


def _nms_into(boxes, keep, thresh, rotated):
    """Reference contract (iou3d_nms.py:86-89): `boxes` already sorted by score, `keep` a HOST LongTensor that
    receives the kept positions, returns their number."""
    from .operators.iou3d_nms import _nms

    order = torch.arange(boxes.shape[0], 0, -1, device=boxes.device, dtype=torch.float32)  # keeps the given order
    kept, _ = _nms(boxes, order, thresh, None, rotated)
    n = int(kept.numel())
    keep[:n] = kept.to(keep.device)
    return n


def nms_gpu(boxes, keep, thresh):
    return _nms_into(boxes, keep, thresh, True)


def nms_normal_gpu(boxes, keep, thresh):
    return _nms_into(boxes, keep, thresh, False)


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        def _not_in_scope(*args, **kwargs):
            raise NotImplementedError("efg._C.%s is outside the ConQueR / CenterPoint hot path this package "
                                      "accelerates (SURVEY.md section 2.1 #22-23)" % name)

        _not_in_scope.__name__ = name
        return _not_in_scope
    raise AttributeError("module 'efg._C' has no attribute %r" % name)
