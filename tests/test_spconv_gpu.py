"""GPU parity: sparse-conv geometry (bit-exact rulebooks) and arithmetic (fp32, 1e-4 relative to
the double-accumulating oracle) through the C ABI; plus dense-equivalence and full-size properties."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_oracle_spconv import CONVS, random_sparse

pytestmark = pytest.mark.gpu


def _tensor(dev, idx, feat, batch, shape):
    import efg_amd.spconv as spconv

    return spconv.SparseConvTensor(torch.from_numpy(feat).to(dev), torch.from_numpy(idx).to(dev), shape, batch)


@pytest.mark.parametrize("ks,st,pd", CONVS)
@pytest.mark.parametrize("cin,cout", [(5, 16), (32, 64), (128, 128)])
def test_regular_conv_vs_oracle(dev, oracle_mod, ks, st, pd, cin, cout):
    import efg_amd.spconv as spconv

    rng = np.random.default_rng(cin * 7 + cout)
    batch, shape = 2, (9, 20, 24)
    idx, feat = random_sparse(rng, batch, shape, 1500, cin)
    conv = spconv.SparseConv3d(cin, cout, ks, st, padding=pd, bias=False).to(dev)
    x = _tensor(dev, idx, feat, batch, shape)
    x.features.requires_grad_(True)
    y = conv(x)
    w = conv.weight.detach().cpu().numpy().reshape(cout, -1, cin)
    out_idx, oshape = oracle_mod.spconv_out_indices(idx, batch, shape, ks, st, pd)
    nbr = oracle_mod.spconv_rulebook(idx, out_idx, batch, shape, ks, st, pd)
    assert y.spatial_shape == oshape
    assert np.array_equal(y.indices.cpu().numpy(), out_idx)                       # bit-exact sites + order
    rb = conv._rulebook(x)[0]
    assert np.array_equal(rb.nbr.cpu().numpy(), nbr)                              # bit-exact rulebook
    ref = oracle_mod.spconv_forward(feat, w, None, nbr)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    go = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_allclose(x.features.grad.cpu().numpy(), oracle_mod.spconv_dgrad(go, w, nbr, feat.shape[0]),
                               rtol=1e-4, atol=1e-4)
    gw = oracle_mod.spconv_wgrad(feat, go, nbr)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(gw.shape), gw, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("cin,cout,bias", [(16, 16, False), (16, 32, True), (64, 64, False), (256, 256, False),
                                           (6, 16, True), (48, 80, False),
                                           # the narrow layers' kernel (conv_small_kernel<SPC, NT>), forward and dgrad:
                                           # <1,1> / <2,1>, <2,2> / <8,1>, <8,1> / <4,2>, tile kernel / tile kernel
                                           (3, 8, True), (8, 32, False), (32, 16, True), (32, 32, False)])
def test_subm_conv_vs_oracle(dev, oracle_mod, cin, cout, bias):
    import efg_amd.spconv as spconv

    rng = np.random.default_rng(cin + cout)
    batch, shape = 2, (7, 24, 24)
    idx, feat = random_sparse(rng, batch, shape, 2000, cin)
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=bias, indice_key="k").to(dev)
    x = _tensor(dev, idx, feat, batch, shape)
    x.features.requires_grad_(True)
    y = conv(x)
    assert y.indices is x.indices  # same sites, same row order
    w = conv.weight.detach().cpu().numpy().reshape(cout, 27, cin)
    b = conv.bias.detach().cpu().numpy() if bias else None
    nbr = oracle_mod.spconv_rulebook(idx, idx, batch, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert np.array_equal(x.indice_dict[("subm", "k", (3, 3, 3), tuple(shape))].nbr.cpu().numpy(), nbr)
    ref = oracle_mod.spconv_forward(feat, w, b, nbr)
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    go = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_allclose(x.features.grad.cpu().numpy(), oracle_mod.spconv_dgrad(go, w, nbr, feat.shape[0]),
                               rtol=1e-4, atol=1e-4)
    gw = oracle_mod.spconv_wgrad(feat, go, nbr)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(gw.shape), gw, rtol=1e-4, atol=2e-4)
    if bias:
        np.testing.assert_allclose(conv.bias.grad.cpu().numpy(), go.sum(0), rtol=1e-4, atol=1e-4)


def test_dense_roundtrip_and_grad(dev, oracle_mod):
    rng = np.random.default_rng(5)
    batch, shape, c = 3, (5, 37, 41), 70
    idx, feat = random_sparse(rng, batch, shape, 3000, c)
    x = _tensor(dev, idx, feat, batch, shape)
    x.features.requires_grad_(True)
    d = x.dense()
    assert d.shape == (batch, c, *shape)
    assert np.array_equal(d.detach().cpu().numpy(), oracle_mod.sparse_to_dense(feat, idx, batch, shape))
    g = torch.randn_like(d)
    d.backward(g)
    ref = g.cpu()[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy()
    assert np.array_equal(x.features.grad.cpu().numpy(), ref)


def test_sequential_block_alignment(dev):
    """main and shortcut strided convs over the same input must be row-aligned (sparse_net.py:155-165)."""
    import efg_amd.spconv as spconv

    rng = np.random.default_rng(8)
    idx, feat = random_sparse(rng, 2, (9, 30, 30), 2500, 16)
    x = _tensor(dev, idx, feat, 2, (9, 30, 30))
    main = spconv.SparseSequential(spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False), torch.nn.BatchNorm1d(32),
                                   torch.nn.ReLU(), spconv.SubMConv3d(32, 32, 3, padding=1, bias=False,
                                                                      indice_key="r")).to(dev)
    short = spconv.SparseSequential(spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
                                    torch.nn.BatchNorm1d(32)).to(dev)
    a, b = main(x), short(x)
    assert torch.equal(a.indices, b.indices)
    lin = ((a.indices[:, 0].long() * 5 + a.indices[:, 1]) * 15 + a.indices[:, 2]) * 15 + a.indices[:, 3]
    assert (lin[1:] > lin[:-1]).all()  # canonical order
    (a.features + b.features).sum().backward()


def test_full_size_properties(dev):
    """180k-point scene through the ConQueR stem geometry: dense-free invariants."""
    import efg_amd.spconv as spconv
    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.operators import voxelize_batch

    pts = [torch.from_numpy(make_scene(2000 + i, n_points=180000)[0]).to(dev) for i in range(2)]
    vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
    x = spconv.SparseConvTensor(vox["voxel_mean"], vox["coordinates"], [41, 1504, 1504], 2)
    conv = spconv.SparseConv3d(5, 16, 3, 2, padding=1, bias=False).to(dev)
    sub = spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key="s").to(dev)
    y = conv(x)
    assert y.spatial_shape == [21, 752, 752]
    ind = y.indices.long()
    lin = ((ind[:, 0] * 21 + ind[:, 1]) * 752 + ind[:, 2]) * 752 + ind[:, 3]
    assert (lin[1:] > lin[:-1]).all()
    # every input site maps (k = centre-ish) into exactly the outputs that list it: pairs are consistent
    rb = conv._rulebook(x)[0]
    nbr = rb.nbr
    valid = nbr >= 0
    assert int(valid.sum()) == int((rb.rnbr >= 0).sum())           # nbr and its transpose hold the same pairs
    assert valid.any(0).all()                                        # no output site without an input
    # linearity of the arithmetic
    z1 = sub(y).features
    z2 = sub(y.replace_feature(y.features * 2.0)).features
    torch.testing.assert_close(z2, z1 * 2.0, rtol=1e-5, atol=1e-6)
    # a delta feature propagates only to its neighbours
    f = torch.zeros_like(y.features)
    f[12345, 3] = 1.0
    z = sub(y.replace_feature(f)).features
    touched = torch.nonzero(z.abs().sum(1) > 0).flatten()
    centre = y.indices[12345]
    d = (y.indices[touched] - centre).abs()
    assert (d[:, 0] == 0).all() and (d[:, 1:] <= 1).all()


def test_dense_bev_matches_dense_view(dev):
    """dense_bev() == dense().view(N, C*D, H, W) (sparse_net.py:304-306), stored channels-last; same gradients."""
    rng = np.random.default_rng(11)
    batch, shape, c = 2, (6, 47, 53), 40
    idx, feat = random_sparse(rng, batch, shape, 4000, c)
    x = _tensor(dev, idx, feat, batch, shape)
    x.features.requires_grad_(True)
    bev = x.dense_bev()
    ref = x.dense().view(batch, c * shape[0], shape[1], shape[2])
    assert bev.shape == ref.shape and bev.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(bev, ref)
    g = torch.randn_like(ref)
    (g1,) = torch.autograd.grad(bev, x.features, g, retain_graph=True)
    (g2,) = torch.autograd.grad(ref, x.features, g)
    assert torch.equal(g1, g2)


def test_packed_weight_cache_follows_every_weight_update(dev, monkeypatch):
    """The MFMA-order copies of the weights are cached on the parameters and refreshed in ONE launch after an optimizer
    step (spconv/core.py `_PackRegistry`).  Whatever changes the weights -- the fused AdamW (which does not move
    `_version`), an in-place autograd op, a write through `.data` followed by `weights_updated()` -- the next forward and
    backward must equal the uncached path bit for bit."""
    import efg_amd.spconv as spconv
    from efg_amd.spconv import core

    rng = np.random.default_rng(11)
    batch, shape = 2, (7, 24, 24)
    idx, feat = random_sparse(rng, batch, shape, 3000, 16)
    c1 = spconv.SubMConv3d(16, 64, 3, padding=1, bias=False, indice_key="k").to(dev)
    c2 = spconv.SparseConv3d(64, 64, 3, stride=2, padding=1, bias=False).to(dev)
    opt = torch.optim.AdamW(list(c1.parameters()) + list(c2.parameters()), lr=0.05, fused=True)
    refreshes = []
    real = core._PackRegistry.refresh_all
    mine = {id(c1.weight), id(c2.weight)}

    def counted(self, d):
        real(self, d)
        refreshes.append(sum(1 for k in self.keys if k[0] in mine))   # (other tests' layers may still be alive)

    monkeypatch.setattr(core._PackRegistry, "refresh_all", counted)

    def run():
        x = _tensor(dev, idx, feat, batch, shape)
        x.features.requires_grad_(True)
        y = c2(c1(x)).features
        (y * y).sum().backward()
        out = (y.detach().clone(), x.features.grad.clone(), c1.weight.grad.clone(), c2.weight.grad.clone())
        c1.weight.grad = c2.weight.grad = None
        return out

    def check(what):
        got = run()
        monkeypatch.setattr(core, "_PACK_CACHE_ON", False)
        ref = run()
        monkeypatch.setattr(core, "_PACK_CACHE_ON", True)
        for g, r in zip(got, ref):
            assert torch.equal(g, r), what

    check("first use")
    for step in range(2):
        out = run()
        c1.weight.grad, c2.weight.grad = out[2], out[3]
        opt.step()
        opt.zero_grad(set_to_none=True)
        n = len(refreshes)
        check("after optimizer step %d" % step)
        assert len(refreshes) == n + 1 and refreshes[-1] == 4, "one refresh launch for all four packed copies"
    with torch.no_grad():
        c1.weight.mul_(1.5)                      # in-place autograd op: `_version` moves
    check("after an in-place update")
    c2.weight.data.mul_(0.5)                     # behind autograd's back: the documented contract is to say so
    spconv.weights_updated()
    check("after a write through .data + weights_updated()")


def test_narrow_layers_run_on_the_small_kernel(dev):
    """efg_spconv_small_ok is the launcher's own rule: the res18 stem (5 -> 16, 16 -> 16, 16 -> 32 and the dgrads 16 -> 16,
    32 -> 16) is covered, 32 -> 32 and the wide layers are not, and the host labels its timings accordingly."""
    import ctypes

    from efg_amd import _lib as L
    from efg_amd.spconv import core

    c16, nt = ctypes.c_int(), ctypes.c_int()
    want = {(5, 16): (1, 1), (16, 16): (1, 1), (16, 32): (1, 2), (32, 16): (2, 1), (3, 8): (1, 1), (8, 32): (1, 2)}
    for (cin, cout), inst in want.items():
        assert L.lib().efg_spconv_small_ok(cin, cout, 27, ctypes.byref(c16), ctypes.byref(nt)) == 1, (cin, cout)
        assert (c16.value, nt.value) == inst
        core._SHAPE_CACHE.clear()
        assert core._tile_kernel_name(cin, cout, 27, 1000, 1000) == "conv_small_kernel<%d,%d>" % inst
    for cin, cout, kvol in [(32, 32, 27), (64, 32, 27), (16, 64, 27), (12, 16, 27), (16, 16, 29)]:
        assert L.lib().efg_spconv_small_ok(cin, cout, kvol, None, None) == 0, (cin, cout, kvol)
    assert core._tile_kernel_name(64, 64, 27, 1000, 1000).startswith("conv_tile_kernel<")


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16), (16, 32), (32, 16), (3, 8), (8, 32)])
@pytest.mark.parametrize("kind", ["subm", "strided", "column"])
def test_small_kernel_bits_equal_tile_kernel(dev, monkeypatch, cin, cout, kind):
    """conv_small_kernel reproduces conv_tile_kernel's sums BIT FOR BIT (same operand placement per MFMA, the tile's active
    offsets dealt to four accumulators like the tile kernel's four split-K waves and added in its order; one accumulator
    below 8 offsets): forward and input gradient with the switch on and off (read per call) are the same bits, so nothing
    downstream -- goldens, tolerances, run-to-run digests -- moved when the narrow layers changed kernels."""
    import efg_amd.spconv as spconv

    rng = np.random.default_rng(cin * 31 + cout)
    batch, shape = 2, (9, 36, 40)
    idx, feat = random_sparse(rng, batch, shape, 9000, cin)
    torch.manual_seed(cin + cout)
    if kind == "subm":
        conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=True, indice_key="k").to(dev)
    elif kind == "strided":
        conv = spconv.SparseConv3d(cin, cout, 3, 2, padding=1, bias=False).to(dev)
    else:   # 3 offsets: the tile kernel does not split those over its waves
        conv = spconv.SparseConv3d(cin, cout, (3, 1, 1), (2, 1, 1), padding=(1, 0, 0), bias=False).to(dev)
    outs = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("EFG_CONV_SMALL", sw)
        x = _tensor(dev, idx, feat, batch, shape)
        x.features.requires_grad_(True)
        y = conv(x)
        go = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(y.features.shape)).astype(np.float32)).to(dev)
        y.features.backward(go)
        outs[sw] = (y.features.detach().clone(), x.features.grad.clone())
        conv.zero_grad()
    assert torch.equal(outs["1"][0], outs["0"][0]), "forward bits differ"
    assert torch.equal(outs["1"][1], outs["0"][1]), "dgrad bits differ"


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128), (128, 256), (16, 64)])
def test_pair_launch_equals_two_launches(dev, oracle_mod, cin, cout):
    """The main + shortcut pair of a residual stage (sparse_net.py:125-165) in one launch per product
    (efg_spconv_tiled_pair_f32 / efg_spconv_wgrad_tiled_pair_f32): the forward n-slices and both weight gradients are what the
    two-launch form computes BIT FOR BIT (without stream-K: the joint item list cuts its shares elsewhere), the joint data
    gradient is the sum of the two to fp32 rounding -- and all of it agrees with the fp64 oracle."""
    import efg_amd.spconv as spconv
    from efg_amd.spconv import core

    rng = np.random.default_rng(cin * 13 + cout)
    batch, shape = 2, (9, 40, 44)
    idx, feat = random_sparse(rng, batch, shape, 12000, cin)
    torch.manual_seed(cin + cout)
    ca = spconv.SparseConv3d(cin, cout, 3, 2, padding=1, bias=False).to(dev)
    cb = spconv.SparseConv3d(cin, cout, 3, 2, padding=1, bias=False).to(dev)
    x = _tensor(dev, idx, feat, batch, shape)
    rb = ca._rulebook(x)[0]
    assert cb._rulebook(x)[0] is rb                       # one geometry for both
    wa = ca.weight.detach().reshape(cout, rb.kvol, cin).contiguous()
    wb = cb.weight.detach().reshape(cout, rb.kvol, cin).contiguous()
    assert core._pair_ok(rb, wa, wb, cin, cout)
    f = x.features
    ya, yb = core._conv_forward(f, wa, None, rb, ca.weight), core._conv_forward(f, wb, None, rb, cb.weight)
    pa, pb = core._conv_forward_pair(f, wa, wb, rb, ca.weight, cb.weight)
    streamk = cin >= 128 and cout >= 128
    if streamk:
        torch.testing.assert_close(pa, ya, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(pb, yb, rtol=1e-5, atol=1e-5)
    else:
        assert torch.equal(pa, ya) and torch.equal(pb, yb), "forward pair differs from the two launches"
    ga = torch.from_numpy(rng.standard_normal(tuple(ya.shape)).astype(np.float32)).to(dev)
    gb = torch.from_numpy(rng.standard_normal(tuple(ya.shape)).astype(np.float32)).to(dev)
    d2 = core._conv_dgrad(ga, wa, rb, ca.weight) + core._conv_dgrad(gb, wb, rb, cb.weight)
    dp = core._conv_dgrad_pair(ga, gb, wa, wb, rb, ca.weight, cb.weight)
    torch.testing.assert_close(dp, d2, rtol=1e-5, atol=2e-5)
    nbr = rb.nbr.cpu().numpy()
    ref = (oracle_mod.spconv_dgrad(ga.cpu().numpy(), wa.cpu().numpy(), nbr, feat.shape[0])
           + oracle_mod.spconv_dgrad(gb.cpu().numpy(), wb.cpu().numpy(), nbr, feat.shape[0]))
    np.testing.assert_allclose(dp.cpu().numpy(), ref, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(pb.cpu().numpy(), oracle_mod.spconv_forward(feat, wb.cpu().numpy(), None, nbr), rtol=1e-4, atol=1e-4)
    wga, wgb = core._conv_wgrad(f, ga, rb), core._conv_wgrad(f, gb, rb)
    wpa, wpb = core._conv_wgrad_pair(f, ga, gb, rb)
    assert torch.equal(wpa, wga) and torch.equal(wpb, wgb), "weight-gradient pair differs from the two launches"


def test_residual_stage_pair_node_matches_module_chain(dev, monkeypatch):
    """SparseBasicResBlock(stride 2) with the pair node (EFG_CONV_PAIR=1, the default) against the chain of single fused
    nodes (EFG_CONV_PAIR=0): outputs, running statistics and every gradient."""
    from efg_amd.modeling.backbones.sparse_net import SparseBasicResBlock

    rng = np.random.default_rng(3)
    batch, shape = 2, (9, 40, 44)
    idx, feat = random_sparse(rng, batch, shape, 12000, 32)
    res = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("EFG_CONV_PAIR", sw)
        torch.manual_seed(11)
        blk = SparseBasicResBlock(32, 64, stride=2, norm="BN1d", activation={"type": "ReLU", "inplace": False}, indice_key="res2").to(dev).train()
        x = _tensor(dev, idx, feat, batch, shape)
        x.features.requires_grad_(True)
        y = blk(x)
        go = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(y.features.shape)).astype(np.float32)).to(dev)
        y.features.backward(go)
        res[sw] = ([y.features.detach().clone(), x.features.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
                   + [b.clone().float() for b in blk.buffers()])
    assert len(res["1"]) == len(res["0"]) and len(res["1"]) > 10
    for a, b in zip(res["1"], res["0"]):
        scale = float(b.abs().max().clamp_min(1e-6))
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, (a.shape, float((a - b).abs().max()), scale)
