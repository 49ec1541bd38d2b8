"""Device-side linear sum assignment (SURVEY.md section 8(f) "GPU matcher") over libefg_hip.so.

Stands in for `scipy.optimize.linear_sum_assignment` at $CQ/modules/matcher.py:89 when the cost matrices are
already on the GPU: same assignment (scipy's algorithm, arithmetic and tie-breaking), no device->host transfer."""
import os

import torch

from .. import _lib


_STATUS = {}  # device -> list of the per-problem status vectors written since the last `take_failures`


def take_failures(device):
    """Device-side count (0-dim tensor) of infeasible assignments (non-finite costs, e.g. after diverged weights) accumulated by
    `linear_sum_assignment_batched` on `device`, reset to zero; None if none was ever recorded.  scipy raises
    ValueError in that situation ($CQ/modules/matcher.py:89); here the count is kept on the device so that the
    training step stays free of host round trips, and the engine reads it together with its loss check
    (efg_amd/engine.py: Trainer._check_anomaly).  Unmatched columns come back as -1 and the loss kernels skip them."""
    pending = _STATUS.pop(torch.device(device), None)
    return None if not pending else torch.cat(pending).ne(0).sum()


def linear_sum_assignment_batched(cost, ng):
    """cost f32 [P, Nq, G] (device), ng int32 [P] (device): columns >= ng[p] of problem p are padding.
    Returns query_of_gt int64 [P, G] (device): the query matched to each GT column, -1 where padded.
    The reference's (row_ind, col_ind) for problem p are {(query_of_gt[p, g], g) : g < ng[p]}."""
    _lib.require_gpu(cost, ng)
    assert cost.dim() == 3 and ng.dim() == 1 and ng.shape[0] == cost.shape[0] and ng.dtype == torch.int32
    cost = cost.contiguous().float()
    p, nq, g = cost.shape
    out = torch.empty((p, g), dtype=torch.int64, device=cost.device)
    status = torch.empty((p,), dtype=torch.int32, device=cost.device)
    _lib.check(_lib.lib().efg_lsap_f32(_lib.ptr(cost), p, nq, g, _lib.ptr(ng.contiguous()), _lib.ptr(out),
                                       _lib.ptr(status), _lib.stream()))
    pending = _STATUS.setdefault(cost.device, [])
    pending.append(status)  # no kernels here: the vectors are folded when the engine asks (every N steps)
    if len(pending) > 4096:  # nobody is asking (a bare loop without the Trainer): keep the list bounded
        del pending[:-64]
    if os.environ.get("EFG_CHECK_LSAP", "0") == "1" and bool(status.any()):  # debugging aid: drains the stream
        raise ValueError("linear_sum_assignment: cost matrix is infeasible (non-finite entries)")
    return out
