"""Rotated BEV IoU / 3-D IoU / NMS -- host mirror of efg/operators/iou3d_nms.py:19-128 over libefg_hip.so.

Same function names, argument meaning and return values as the reference wrappers.  Differences that do not
change results: the 3-D IoU composition (iou3d_nms.py:54-87) is fused into the pair kernel (same fp32 operation
order), and NMS suppression runs on the GPU -- the only device->host transfer is the 4-byte kept count, where
the reference copies the N x N/64 mask to the host (iou3d_nms.cpp:96-100).
"""
import torch

from .. import _lib

_MODE = {"overlap": 0, "iou": 1, "iou3d": 2}


def _check_boxes(*boxes):
    for b in boxes:
        assert b.dim() == 2 and b.shape[1] == 7, "boxes must be (N, 7) [x, y, z, dx, dy, dz, heading]"
    _lib.require_gpu(*boxes)


def _pairwise(boxes_a, boxes_b, mode):
    _check_boxes(boxes_a, boxes_b)
    a, b = boxes_a.contiguous().float(), boxes_b.contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().efg_boxes_bev_f32(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0], _MODE[mode],
                                            _lib.ptr(out), _lib.stream()))
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N,7) x (M,7) -> (N,M) rotated-rectangle intersection AREA (efg::boxes_overlap_bev_gpu)."""
    return _pairwise(boxes_a, boxes_b, "overlap")


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7) x (M,7) -> (N,M) rotated BEV IoU (iou3d_nms.py:38-51)."""
    return _pairwise(boxes_a, boxes_b, "iou")


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7) x (M,7) -> (N,M) 3-D IoU = BEV overlap x height overlap / union volume (iou3d_nms.py:54-87)."""
    return _pairwise(boxes_a, boxes_b, "iou3d")


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """The reference's CPU entry point (iou3d_nms.py:19-35).  This package has no CPU compute path."""
    raise RuntimeError("efg_amd: boxes_bev_iou_cpu has no CPU implementation; move the boxes to the GPU and "
                       "call boxes_iou_bev")


def _nms(boxes, scores, thresh, pre_maxsize, rotated):
    _check_boxes(boxes)
    _lib.require_gpu(scores)
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].contiguous().float()
    n = b.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    num = torch.empty((1,), dtype=torch.int32, device=b.device)
    ws_bytes = _lib.lib().efg_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=b.device)
    _lib.check(_lib.lib().efg_nms_f32(_lib.ptr(b), n, float(thresh), 1 if rotated else 0, _lib.ptr(keep),
                                      _lib.ptr(num), _lib.ptr(ws), ws_bytes, _lib.stream()))
    num_out = int(num.item())
    return order[keep[:num_out]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated NMS (iou3d_nms.py:90-106): returns (indices into ``boxes`` of the kept ones, None)."""
    return _nms(boxes, scores, thresh, pre_maxsize, True)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned BEV NMS ignoring the heading (iou3d_nms.py:109-124)."""
    return _nms(boxes, scores, thresh, None, False)
