"""CPU: pin the oracle's voxelizers against the reference's golden vectors (tests/golden/
voxelize_*.npz, produced by scripts/make_golden.py from voxelization_cpu.cpp + the numba twin)
and, when the in-place reference build oracle/_ref is present, against the reference itself."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "voxelize_*.npz")))


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_golden(oracle_mod, case):
    g = golden(case)
    v, c, n = oracle_mod.hard_voxelize(g["points"], g["voxel_size"], g["coors_range"], int(g["max_points"]),
                                       int(g["max_voxels"]))
    assert np.array_equal(v, g["voxels"])
    assert np.array_equal(c, g["coors"])
    assert np.array_equal(n, g["num_points_per_voxel"])
    d = oracle_mod.dynamic_voxelize(g["points"], g["voxel_size"], g["coors_range"])
    assert np.array_equal(d, g["dynamic_coors"])


def test_break_case_really_breaks():
    g = golden("voxelize_break_6k.npz")
    assert g["voxels"].shape[0] == int(g["max_voxels"])  # the cap (and its `break`) is exercised
    assert g["num_points_per_voxel"].max() == int(g["max_points"])


def test_oracle_matches_reference_build(oracle_mod):
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene

    for seed, n, mp, mv in [(11, 30000, 5, 120000), (12, 30000, 2, 5000), (13, 500, 1, 10)]:
        p, _, _ = make_scene(seed, n_points=n)
        a = oracle_mod.hard_voxelize(p, VOXEL_SIZE, PC_RANGE, mp, mv)
        b = oracle_mod.hard_voxelize(p, VOXEL_SIZE, PC_RANGE, mp, mv, use_ref=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert np.array_equal(oracle_mod.dynamic_voxelize(p, VOXEL_SIZE, PC_RANGE),
                              oracle_mod.dynamic_voxelize(p, VOXEL_SIZE, PC_RANGE, use_ref=True))


def test_voxel_mean(oracle_mod):
    g = golden("voxelize_cfg0_16k.npz")
    m = oracle_mod.voxel_mean(g["voxels"], g["num_points_per_voxel"])
    ref = g["voxels"].sum(1) / g["num_points_per_voxel"][:, None].astype(np.float32)
    np.testing.assert_allclose(m, ref, rtol=1e-6, atol=1e-6)


def test_empty_and_all_outside(oracle_mod):
    vs, cr = (0.5, 0.5, 0.5), (0, 0, 0, 4, 4, 4)
    v, c, n = oracle_mod.hard_voxelize(np.zeros((0, 4), np.float32), vs, cr, 3, 10)
    assert v.shape == (0, 3, 4) and c.shape == (0, 3) and n.shape == (0,)
    pts = np.full((7, 4), -3.0, np.float32)
    v, c, n = oracle_mod.hard_voxelize(pts, vs, cr, 3, 10)
    assert v.shape[0] == 0
    assert (oracle_mod.dynamic_voxelize(pts, vs, cr) == -1).all()


@pytest.mark.parametrize("reduce", ["max", "mean", "sum"])
def test_scatter_oracle_pinned_to_reference_cpu_grouping(oracle_mod, reduce):
    """The scatter restatement (oracle_scatter_forward, after scatter_points_cuda.cu:209-290) against the REFERENCE's
    own CPU grouping dynamic_point_to_voxel_cpu (scatter_points_cpu.cpp:62-119, compiled in place into oracle/_ref):
    same voxel set, same counts, same membership; max exact, sum / mean vs an fp64 sum of the reference's groups."""
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(5)
    n, c = 30000, 5
    pts = np.empty((n, c), np.float32)
    pts[:, 0:2] = rng.uniform(-10, 10, (n, 2))
    pts[:, 2] = rng.uniform(-1, 1, n)
    pts[:, 3:] = rng.standard_normal((n, c - 3))
    vs, cr = [0.25, 0.25, 0.5], [-10, -10, -1, 10, 10, 1]
    coors = oracle_mod.dynamic_voxelize(pts, vs, cr, use_ref=True)
    keep = coors[:, 0] >= 0   # fp32 rounding puts a few points ON the upper range; the reference grouping takes no -1 rows
    pts, coors = np.ascontiguousarray(pts[keep]), np.ascontiguousarray(coors[keep])
    groups, vc, npv = oracle_mod.ref_dynamic_point_to_voxel(pts, coors, vs, cr)
    order = np.lexsort((vc[:, 2], vc[:, 1], vc[:, 0]))
    groups, vc, npv = groups[order], vc[order], npv[order]
    vf, ovc, p2v, cnt = oracle_mod.scatter_forward(pts, coors, reduce)
    assert np.array_equal(ovc, vc)
    if reduce == "mean":   # the CUDA reference counts only for the mean (scatter_points_cuda.cu:101-133)
        assert np.array_equal(cnt, npv)
    assert np.array_equal(vc[p2v], coors)
    valid = np.arange(groups.shape[1])[None, :, None] < npv[:, None, None]
    if reduce == "max":
        assert np.array_equal(vf, np.where(valid, groups, -np.inf).max(1).astype(np.float32))
    else:
        exp = groups.astype(np.float64).sum(1)
        if reduce == "mean":
            exp = exp / npv[:, None]
        np.testing.assert_allclose(vf, exp, rtol=1e-5, atol=1e-5)
