"""GPU parity: dynamic scatter HIP path vs the oracle restatement of scatter_points_cuda.cu.
Indices / counts / max are bit-exact; sum and mean use fp32 atomics like the reference (its own
docstring quotes ~5e-7 CPU/GPU differences, efg/operators/scatter_points.py:59-60) -> 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(n, c, seed, frac_invalid=0.05):
    rng = np.random.default_rng(seed)
    coors = rng.integers(0, [6, 40, 40], size=(n, 3)).astype(np.int32)
    bad = rng.random(n) < frac_invalid
    coors[bad] = -1
    feats = rng.standard_normal((n, c)).astype(np.float32)
    return feats, coors


@pytest.mark.parametrize("reduce", ["max", "mean", "sum"])
@pytest.mark.parametrize("n,c", [(20000, 5), (777, 9), (1, 4)])
def test_forward_backward_vs_oracle(dev, oracle_mod, reduce, n, c):
    from efg_amd.operators.scatter_points import dynamic_scatter

    feats, coors = _cloud(n, c, seed=n + c)
    evf, evc, ep2v, ecnt = oracle_mod.scatter_forward(feats, coors, reduce)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    vf, vc = dynamic_scatter(f, torch.from_numpy(coors).to(dev), reduce)
    assert np.array_equal(vc.cpu().numpy(), evc)
    if reduce == "max":
        assert np.array_equal(vf.detach().cpu().numpy(), evf)
    else:
        np.testing.assert_allclose(vf.detach().cpu().numpy(), evf, rtol=1e-5, atol=1e-5)
    gvox = np.random.default_rng(1).standard_normal(evf.shape).astype(np.float32)
    vf.backward(torch.from_numpy(gvox).to(dev))
    eg = oracle_mod.scatter_backward(gvox, feats, evf, ep2v, ecnt, reduce)
    if reduce == "mean":
        np.testing.assert_allclose(f.grad.cpu().numpy(), eg, rtol=1e-6, atol=1e-7)
    elif reduce == "sum":
        assert np.array_equal(f.grad.cpu().numpy(), eg)
    else:
        # ties at the max are resolved to the lowest point index on both sides only if the
        # forward maxima are bit-identical, which they are (max is exact)
        assert np.array_equal(f.grad.cpu().numpy(), eg)


def test_map_and_count_bit_exact(dev, oracle_mod):
    from efg_amd.operators.scatter_points import dynamic_point_to_voxel_forward

    feats, coors = _cloud(50000, 4, seed=9)
    evf, evc, ep2v, ecnt = oracle_mod.scatter_forward(feats, coors, "mean")
    vf, vc, p2v, cnt = dynamic_point_to_voxel_forward(torch.from_numpy(feats).to(dev),
                                                      torch.from_numpy(coors).to(dev), "mean")
    assert np.array_equal(p2v.cpu().numpy(), ep2v)
    assert np.array_equal(cnt.cpu().numpy(), ecnt)
    assert np.array_equal(vc.cpu().numpy(), evc)
    # sorted, unique keys (the reference's argsort order)
    key = (evc[:, 0].astype(np.int64) * 40 + evc[:, 1]) * 40 + evc[:, 2]
    assert (np.diff(key) > 0).all()


def test_all_invalid_and_empty(dev):
    from efg_amd.operators.scatter_points import dynamic_point_to_voxel_forward

    f = torch.randn(10, 3, device=dev)
    c = torch.full((10, 3), -1, dtype=torch.int32, device=dev)
    vf, vc, p2v, cnt = dynamic_point_to_voxel_forward(f, c, "max")
    assert vf.shape == (0, 3) and (p2v == -1).all()
    vf, vc, p2v, cnt = dynamic_point_to_voxel_forward(f[:0], c[:0], "sum")
    assert vf.shape == (0, 3)
    with pytest.raises(RuntimeError):
        dynamic_point_to_voxel_forward(f, c, "median")


def test_dynamic_scatter_module_batched(dev, oracle_mod):
    from efg_amd.operators import DynamicScatter

    feats, coors = _cloud(6000, 4, seed=4, frac_invalid=0.0)
    b = (np.arange(6000) // 2000).astype(np.int32)
    coors4 = np.concatenate([b[:, None], coors], 1)
    mod = DynamicScatter([0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1], average_points=True)
    vf, vc = mod(torch.from_numpy(feats).to(dev), torch.from_numpy(coors4).to(dev))
    rows = 0
    for i in range(3):
        evf, evc, _, _ = oracle_mod.scatter_forward(feats[b == i], coors[b == i], "mean")
        sl = slice(rows, rows + evf.shape[0])
        assert (vc[sl, 0] == i).all() and np.array_equal(vc[sl, 1:].cpu().numpy(), evc)
        np.testing.assert_allclose(vf[sl].cpu().numpy(), evf, rtol=1e-5, atol=1e-5)
        rows += evf.shape[0]
    assert vf.shape[0] == rows


@pytest.mark.parametrize("reduce", ["max", "mean", "sum"])
@pytest.mark.parametrize("n,c", [(30000, 6), (4097, 33)])
def test_forward_backward_vs_torch_unique_scatter_reduce(dev, reduce, n, c):
    """Independent of oracle/: the semantics of scatter_points_cuda.cu:209-352 written with stock PyTorch ops in
    fp64 -- voxel id = rank of the row-major key (dims = coors.max(0) + 1) among the distinct valid keys
    (`torch.unique`, ascending), features reduced per voxel with `scatter_reduce`, gradients by autograd through it
    (no ties in random data, so amax's tie rule does not matter)."""
    from efg_amd.operators.scatter_points import dynamic_point_to_voxel_forward, dynamic_scatter

    feats, coors = _cloud(n, c, seed=3 * n + c)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    cg = torch.from_numpy(coors).to(dev)
    vf, vc = dynamic_scatter(f, cg, reduce)
    _, _, p2v, cnt = dynamic_point_to_voxel_forward(f.detach(), cg, reduce)
    ct = torch.from_numpy(coors).long()
    valid = (ct >= 0).all(1)
    dims = ct.max(0)[0] + 1
    key = (ct[:, 0] * dims[1] + ct[:, 1]) * dims[2] + ct[:, 2]
    uniq, inv = torch.unique(key[valid], return_inverse=True)  # sorted ascending
    m = uniq.numel()
    ref_map = torch.full((n,), -1, dtype=torch.int64)
    ref_map[valid] = inv
    assert np.array_equal(p2v.cpu().numpy(), ref_map.numpy().astype(np.int32))
    ref_coors = torch.stack([uniq // (dims[1] * dims[2]), (uniq // dims[2]) % dims[1], uniq % dims[2]], 1)
    assert np.array_equal(vc.cpu().numpy(), ref_coors.numpy().astype(np.int32))
    x = torch.from_numpy(feats).double().requires_grad_(True)
    red = {"max": "amax", "mean": "mean", "sum": "sum"}[reduce]
    ref = torch.zeros(m, c, dtype=torch.float64).scatter_reduce(0, inv[:, None].expand(-1, c), x[valid], red,
                                                                include_self=False)
    np.testing.assert_allclose(vf.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    if reduce == "mean":
        assert np.array_equal(cnt.cpu().numpy(), torch.bincount(inv, minlength=m).numpy().astype(np.int32))
    g = np.random.default_rng(2).standard_normal((m, c)).astype(np.float32)
    vf.backward(torch.from_numpy(g).to(dev))
    ref.backward(torch.from_numpy(g).double())
    np.testing.assert_allclose(f.grad.cpu().numpy(), x.grad.numpy(), rtol=1e-5, atol=1e-6)


def _ref_groups(oracle_mod, n, c, seed):
    """A cloud voxelized by the reference's own dynamic_voxelize_cpu and grouped by the reference's own
    dynamic_point_to_voxel_cpu (scatter_points_cpu.cpp:62-119; both compiled in place into oracle/_ref): per voxel,
    ALL its points.  Sorted by linearised key, the order scatter_points_cuda.cu:209-290 and the HIP path emit."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n, c), np.float32)
    pts[:, 0:2] = rng.uniform(-10, 10, (n, 2))
    pts[:, 2] = rng.uniform(-1, 1, n)
    pts[:, 3:] = rng.standard_normal((n, c - 3))
    vs, cr = [0.25, 0.25, 0.5], [-10, -10, -1, 10, 10, 1]
    coors = oracle_mod.dynamic_voxelize(pts, vs, cr, use_ref=True)
    keep = coors[:, 0] >= 0
    pts, coors = np.ascontiguousarray(pts[keep]), np.ascontiguousarray(coors[keep])
    groups, vc, npv = oracle_mod.ref_dynamic_point_to_voxel(pts, coors, vs, cr)
    order = np.lexsort((vc[:, 2], vc[:, 1], vc[:, 0]))
    return pts, coors, groups[order], vc[order], npv[order]


@pytest.mark.parametrize("reduce", ["max", "mean", "sum"])
def test_reduce_pinned_to_reference_cpu_grouping(dev, oracle_mod, reduce):
    """a3 pinned to the REFERENCE: voxel set, per-voxel counts and the reduce over exactly the reference's point groups.
    max is exact; sum / mean are compared with an fp64 sum of the reference's groups at 1e-5 (fp32 atomics, any order)."""
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    from efg_amd.operators.scatter_points import dynamic_point_to_voxel_forward

    pts, coors, groups, vc, npv = _ref_groups(oracle_mod, 40000, 6, seed=21)
    vf, hvc, p2v, cnt = dynamic_point_to_voxel_forward(torch.from_numpy(pts).to(dev), torch.from_numpy(coors).to(dev),
                                                       reduce)
    assert np.array_equal(hvc.cpu().numpy(), vc)
    if reduce == "mean":   # the CUDA reference counts only for the mean (scatter_points_cuda.cu:101-133)
        assert np.array_equal(cnt.cpu().numpy(), npv)
    # every point sits in the voxel the reference put it in
    assert np.array_equal(vc[p2v.cpu().numpy()], coors)
    valid = np.arange(groups.shape[1])[None, :, None] < npv[:, None, None]
    if reduce == "max":
        exp = np.where(valid, groups, -np.inf).max(1)
        assert np.array_equal(vf.cpu().numpy(), exp.astype(np.float32))
    else:
        exp = groups.astype(np.float64).sum(1)   # padding rows are zero
        if reduce == "mean":
            exp = exp / npv[:, None]
        np.testing.assert_allclose(vf.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_sum_and_mean_do_not_depend_on_the_point_order(dev, reduce):
    """Sums are accumulated as exact 64-bit fixed-point integers (csrc/scatter.hip): the SAME points in any order --
    a different arrival order of the atomics -- give the same bits, and the result is the correctly rounded fp64 sum.
    (With float atomics two runs of one call already differ in the last bits.)"""
    from efg_amd.operators.scatter_points import dynamic_scatter

    rng = np.random.default_rng(3)
    n, c = 60000, 7
    coors = rng.integers(0, [4, 12, 12], size=(n, 3)).astype(np.int32)     # ~100 points per voxel: long sums
    feats = (rng.standard_normal((n, c)) * np.exp(rng.uniform(-12, 6, (n, 1)))).astype(np.float32)   # 8 decades of magnitude
    outs = []
    for trial in range(3):
        perm = rng.permutation(n) if trial else np.arange(n)
        vf, vc = dynamic_scatter(torch.from_numpy(feats[perm]).to(dev), torch.from_numpy(coors[perm]).to(dev), reduce)
        outs.append((vf.cpu().numpy(), vc.cpu().numpy()))
    for vf, vc in outs[1:]:
        assert np.array_equal(vc, outs[0][1])
        assert np.array_equal(vf, outs[0][0]), "voxel %s of a permuted cloud differs in its bits" % reduce
    # against the exact sum: key -> float64 accumulation
    key = (coors[:, 0].astype(np.int64) * 12 + coors[:, 1]) * 12 + coors[:, 2]
    uniq, inv = np.unique(key, return_inverse=True)
    ref = np.zeros((len(uniq), c))
    np.add.at(ref, inv, feats.astype(np.float64))
    if reduce == "mean":
        cnt = np.bincount(inv).astype(np.float32)
        ref = ref.astype(np.float32) / cnt[:, None]
    else:
        ref = ref.astype(np.float32)
    got = outs[0][0]
    ulp = np.abs(np.spacing(ref))
    assert np.all(np.abs(got - ref) <= 1.5 * ulp), "not the correctly rounded sum"
