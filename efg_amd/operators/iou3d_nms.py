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


def _nms_segmented(boxes_sorted, segment, thresh, rotated):
    """boxes_sorted [n, 7] (segment-major, score-descending inside a segment), segment int32 [n] ascending ->
    (keep int64 [n] device buffer, kept count int32 [1] device)."""
    _check_boxes(boxes_sorted)
    b = boxes_sorted.contiguous().float()
    n = b.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    num = torch.empty((1,), dtype=torch.int32, device=b.device)
    ws_bytes = _lib.lib().efg_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=b.device)
    seg = segment.contiguous().to(torch.int32)
    _lib.check(_lib.lib().efg_nms_segmented_f32(_lib.ptr(b), _lib.ptr(seg), n, float(thresh), 1 if rotated else 0,
                                                _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws_bytes, _lib.stream()))
    return keep, num


def nms_gpu_batched(boxes, scores, thresh, score_thresh=None, rotated=True, valid=None):
    """`nms_gpu` over S independent sets of M boxes each in one launch (ours; the reference loops in Python).

    boxes [S, M, 7], scores [S, M].  Returns (set_id int64 [K], index int64 [K], counts int64 [S]): the kept boxes set
    by set, inside a set in the order `nms_gpu(boxes[s][valid], scores[s][valid])` returns them, `index` pointing
    into the set's M boxes; boxes with score < score_thresh are dropped, and so are the entries where `valid`
    (bool [S, M], e.g. padding of ragged sets) is False.  One device->host read (K)."""
    s, m = scores.shape
    if valid is not None:
        scores = torch.where(valid, scores, torch.full_like(scores, -float("inf")))   # pads sort behind everything
    order = scores.sort(dim=1, descending=True)[1]                       # per set, as nms_gpu sorts (iou3d_nms.py:98)
    sorted_boxes = torch.gather(boxes, 1, order[..., None].expand(-1, -1, 7)).reshape(s * m, 7)
    set_id = torch.arange(s, device=scores.device).repeat_interleave(m)
    keep, num = _nms_segmented(sorted_boxes, set_id, thresh, rotated)
    n = s * m
    live = torch.arange(n, device=scores.device) < num                    # the first `num` entries of `keep` are valid
    keep = keep.clamp(0, n - 1)
    if score_thresh is not None or valid is not None:
        # a box below the score threshold (or a pad) sorts behind every valid box of its set, so it cannot have
        # suppressed one; it only has to be dropped from the result
        floor = -float("inf") if score_thresh is None else score_thresh
        sorted_scores = torch.gather(scores, 1, order).reshape(-1)[keep]
        live = live & (sorted_scores >= floor) & (sorted_scores > -float("inf"))
    kept = keep[live]                                                     # the one compaction (device->host size)
    kept_set = torch.div(kept, m, rounding_mode="floor")
    index = order.reshape(-1)[kept]
    counts = torch.zeros(s, dtype=torch.int64, device=scores.device).index_add_(0, kept_set, torch.ones_like(kept_set))
    return kept_set, index, counts


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated NMS (iou3d_nms.py:90-106): returns (indices into ``boxes`` of the kept ones, None)."""
    return _nms(boxes, scores, thresh, pre_maxsize, True)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned BEV NMS ignoring the heading (iou3d_nms.py:109-124)."""
    return _nms(boxes, scores, thresh, None, False)
