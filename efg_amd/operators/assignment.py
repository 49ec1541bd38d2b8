"""Device-side linear sum assignment (SURVEY.md section 8(f) "GPU matcher") over libefg_hip.so.

Stands in for `scipy.optimize.linear_sum_assignment` at $CQ/modules/matcher.py:89 when the cost matrices are
already on the GPU: same assignment (scipy's algorithm, arithmetic and tie-breaking), no device->host transfer."""
import os

import torch

from .. import _lib


def linear_sum_assignment_batched(cost, ng):
    """cost f32 [P, Nq, G] (device), ng int32 [P] (device): columns >= ng[p] of problem p are padding.
    Returns query_of_gt int64 [P, G] (device): the query matched to each GT column, -1 where padded.
    The reference's (row_ind, col_ind) for problem p are {(query_of_gt[p, g], g) : g < ng[p]}."""
    _lib.require_gpu(cost, ng)
    assert cost.dim() == 3 and ng.dim() == 1 and ng.shape[0] == cost.shape[0] and ng.dtype == torch.int32
    cost = cost.contiguous().float()
    p, nq, g = cost.shape
    out = torch.empty((p, g), dtype=torch.int64, device=cost.device)
    check = os.environ.get("EFG_CHECK_LSAP", "0") == "1"
    status = torch.empty((p,), dtype=torch.int32, device=cost.device) if check else None
    _lib.check(_lib.lib().efg_lsap_f32(_lib.ptr(cost), p, nq, g, _lib.ptr(ng.contiguous()), _lib.ptr(out),
                                       _lib.ptr(status), _lib.stream()))
    if check and bool(status.any()):  # debugging aid only: this read-back drains the stream
        raise ValueError("linear_sum_assignment: cost matrix is infeasible (non-finite entries)")
    return out
