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
