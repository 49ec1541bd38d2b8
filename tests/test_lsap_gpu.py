"""GPU parity: efg_lsap_f32 vs the oracle (itself pinned to scipy in tests/test_oracle_lsap.py) -- identical
assignments, ties included."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _run(costs, ngs):
    from efg_amd.operators.assignment import linear_sum_assignment_batched
    c = torch.from_numpy(np.stack(costs)).cuda()
    ng = torch.tensor(ngs, dtype=torch.int32).cuda()
    return linear_sum_assignment_batched(c, ng).cpu().numpy()


@pytest.mark.parametrize("nq,g", [(1000, 40), (1000, 1), (300, 157), (64, 64), (5, 9), (1, 1), (200, 199), (17, 400),
                                  (2800, 120)])
def test_random_and_ties(nq, g):
    rng = np.random.default_rng(nq + 31 * g)
    costs = [rng.normal(size=(nq, g)).astype(np.float32) * 3 for _ in range(3)]
    costs += [rng.integers(0, hi, size=(nq, g)).astype(np.float32) for hi in (2, 3, 10)]
    costs += [np.ones((nq, g), np.float32)]
    ngs = [g, max(g // 2, 1), g, g, max(g - 1, 1), g, g]
    got = _run(costs, ngs)
    for c, n, out in zip(costs, ngs, got):
        np.testing.assert_array_equal(out[:n], oracle.lsap(c, ng=n))
        assert (out[n:] == -1).all()


def test_matches_scipy_on_detr_like_costs():
    """Cost structure of the real matcher: focal class cost + L1 + GIoU terms, 6 problems at once."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(0)
    costs, ngs = [], []
    for p in range(6):
        ng = int(rng.integers(20, 60))
        c = np.abs(rng.normal(size=(1000, 64))).astype(np.float32) * 2 - rng.random((1000, 64)).astype(np.float32)
        costs.append(c)
        ngs.append(ng)
    got = _run(costs, ngs)
    for c, n, out in zip(costs, ngs, got):
        i, j = linear_sum_assignment(c[:, :n])
        ref = np.full(n, -1, np.int64)
        ref[j] = i
        np.testing.assert_array_equal(out[:n], ref)


def test_zero_gt_and_status():
    from efg_amd.operators.assignment import linear_sum_assignment_batched
    c = torch.randn(2, 50, 8, device="cuda")
    out = linear_sum_assignment_batched(c, torch.tensor([0, 3], dtype=torch.int32, device="cuda"))
    assert (out[0] == -1).all() and (out[1, :3] >= 0).all() and (out[1, 3:] == -1).all()
    assert len(set(out[1, :3].tolist())) == 3
