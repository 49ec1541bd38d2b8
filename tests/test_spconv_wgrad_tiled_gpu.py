"""GPU: the plan-walking weight gradient (csrc/spconv_wgt.hip) against the table-walking kernel it replaces and an
fp64 sum, at the geometry of a real 180k-point scene (the stem's 16 / 32 channels up to 128 -> 256), plus
run-to-run bit-reproducibility.  The small-shape oracle comparisons of test_spconv_gpu.py go through it as well."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _level(dev, n_down):
    import efg_amd.spconv as spconv
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.operators import voxelize_batch

    pts = [torch.from_numpy(make_scene(2000 + i, n_points=180000)[0]).to(dev) for i in range(2)]
    vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
    x = spconv.SparseConvTensor(vox["voxel_mean"], vox["coordinates"], [41, 1504, 1504], 2)
    cin = 5
    for _ in range(n_down):   # geometry only
        x = spconv.SparseConv3d(cin, 4, 3, 2, padding=1, bias=False).to(dev)(x)
        cin = 4
    return x


@pytest.mark.parametrize("kind,n_down,cin,cout", [("down", 0, 5, 16), ("subm", 1, 16, 16), ("subm", 1, 16, 32), ("down", 1, 32, 64),
                                                  ("subm", 2, 64, 64), ("subm", 3, 128, 128), ("down", 3, 128, 256),
                                                  ("head", 3, 128, 128), ("subm", 2, 32, 32)])
def test_tiled_wgrad_matches_table_kernel(dev, monkeypatch, kind, n_down, cin, cout):
    import efg_amd.spconv as spconv
    from efg_amd.spconv import core

    torch.manual_seed(cin + cout)
    x = _level(dev, n_down)
    if kind == "subm":
        conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k").to(dev)
    elif kind == "down":
        conv = spconv.SparseConv3d(cin, cout, 3, 2, padding=1, bias=False).to(dev)
    else:
        conv = spconv.SparseConv3d(cin, cout, (3, 1, 1), (2, 1, 1), padding=(1, 0, 0), bias=False).to(dev)
    feat = torch.randn(x.features.shape[0], cin, device=dev)
    xin = x.replace_feature(feat)
    y = conv(xin)
    rb = conv._rulebook(xin)[0]
    go = torch.randn_like(y.features)
    assert L_ok(cin, cout, rb.kvol)
    monkeypatch.setenv("EFG_WGRAD_TILED", "1")
    core._WGT_OK.clear()          # (the switch is read once per layer shape)
    g1 = core._conv_wgrad(feat, go, rb)
    g1b = core._conv_wgrad(feat, go, rb)
    assert (cin, cout) in rb._wgrad_sched            # the plan-walking kernel really ran
    monkeypatch.setenv("EFG_WGRAD_TILED", "0")
    core._WGT_OK.clear()
    g0 = core._conv_wgrad(feat, go, rb)
    core._WGT_OK.clear()
    assert torch.equal(g1, g1b)                       # fixed summation order
    # fp64 sum over the table, offset by offset
    nbr = rb.nbr.long()
    ref = torch.zeros(cout, rb.kvol, cin, dtype=torch.float64, device=dev)
    for k in range(rb.kvol):
        o = torch.nonzero(nbr[k] >= 0).flatten()
        ref[:, k, :] = go[o].double().t() @ feat[nbr[k][o]].double()
    scale = float(ref.abs().max())
    e1 = float((g1.double() - ref).abs().max()) / scale
    e0 = float((g0.double() - ref).abs().max()) / scale
    assert e1 < 1e-5 and e0 < 1e-5, (e1, e0)


def L_ok(cin, cout, kvol):
    from efg_amd import _lib as L

    return bool(L.lib().efg_spconv_wgrad_tiled_ok(cin, cout, kvol))


def test_uncovered_shapes_keep_the_table_kernel():
    assert not L_ok(48, 80, 27) and not L_ok(64, 64, 32) and not L_ok(24, 32, 27) and not L_ok(16, 48, 27)
    assert L_ok(5, 16, 27) and L_ok(6, 16, 27) and L_ok(16, 16, 27) and L_ok(16, 32, 27) and L_ok(32, 64, 27) and L_ok(64, 64, 27) and L_ok(256, 256, 3)
