"""CPU: the oracle's linear-sum-assignment restatement against scipy.optimize.linear_sum_assignment ITSELF --
the reference's matcher ($CQ/modules/matcher.py:89) calls scipy, so scipy (1.15.3 in this image) is the real
reference here and the oracle is PINNED to it, including tie-breaking."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

import oracle


def scipy_query_of_gt(cost):
    """cost [nq, ng] -> query matched to each GT column (-1 if none), as the reference consumes (i, j)."""
    i, j = linear_sum_assignment(cost)
    out = np.full(cost.shape[1], -1, np.int64)
    out[j] = i
    return out


CASES = [(1000, 40), (1000, 1), (300, 157), (64, 64), (5, 9), (1, 1), (200, 199), (17, 400)]


@pytest.mark.parametrize("nq,ng", CASES)
def test_random_float_costs(nq, ng):
    rng = np.random.default_rng(nq * 1000 + ng)
    for trial in range(3):
        cost = rng.normal(size=(nq, ng)).astype(np.float32) * 3
        np.testing.assert_array_equal(oracle.lsap(cost), scipy_query_of_gt(cost))


@pytest.mark.parametrize("nq,ng", CASES)
def test_tie_heavy_integer_costs(nq, ng):
    """Small integer costs: many exact ties, so the result depends on scipy's scan order."""
    rng = np.random.default_rng(nq * 7 + ng)
    for hi in (2, 3, 10):
        cost = rng.integers(0, hi, size=(nq, ng)).astype(np.float32)
        np.testing.assert_array_equal(oracle.lsap(cost), scipy_query_of_gt(cost))


def test_constant_and_duplicate_rows():
    for nq, ng in [(50, 50), (80, 13), (13, 80)]:
        cost = np.ones((nq, ng), np.float32)
        np.testing.assert_array_equal(oracle.lsap(cost), scipy_query_of_gt(cost))
    rng = np.random.default_rng(0)
    base = rng.normal(size=(10, 25)).astype(np.float32)
    cost = np.repeat(base, 30, axis=0)  # 300 queries, every row repeated 30 times
    np.testing.assert_array_equal(oracle.lsap(cost), scipy_query_of_gt(cost))


def test_padded_columns_are_ignored():
    rng = np.random.default_rng(4)
    cost = rng.normal(size=(100, 32)).astype(np.float32)
    got = oracle.lsap(cost, ng=20)
    np.testing.assert_array_equal(got, scipy_query_of_gt(cost[:, :20]))


def test_empty():
    assert oracle.lsap(np.zeros((10, 0), np.float32)).shape == (0,)
    assert (oracle.lsap(np.zeros((0, 3), np.float32)) == -1).all()
