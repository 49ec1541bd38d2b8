"""GPU parity of the HIP sparse convolution against an implementation the builder did not write: fp64
`torch.nn.functional.conv3d` on the densified tensor, forward AND both gradients (autograd through the dense conv).

spconv itself is absent from /root/reference, so its results cannot be pinned; what the reference's
backbones rely on is dense equivalence:
  * SubMConv3d(x)   == conv3d(dense(x), W, pad = k // 2) sampled at x's active sites;
  * SparseConv3d(x) == conv3d(dense(x), W, stride, pad) at the sites where conv3d(occupancy, ones) > 0, rows in
    ascending (b, z, y, x) order;
  * gradients == those of the dense conv with the upstream gradient placed at the active output sites.
Covered: every geometry of the ConQueR res18 plan (sparse_net.py:85-95,125-147,273-282) and of CenterPoint's
SpMiddleResNetFHD (:485-524), channel counts 5 ... 256.  Nothing here touches oracle/.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (kernel, stride, padding, subm)
GEOMS = {
    "res18 stem/stage conv k3 s2 p1": ((3, 3, 3), (2, 2, 2), (1, 1, 1), False),
    "res18 subm k3": ((3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    "res18 *_out k(3,1,1) s(2,1,1) p(1,0,0)": ((3, 1, 1), (2, 1, 1), (1, 0, 0), False),
    "centerpoint conv4 k3 s2 p(0,1,1)": ((3, 3, 3), (2, 2, 2), (0, 1, 1), False),
    "centerpoint extra_conv k(3,1,1) s(2,1,1) p0": ((3, 1, 1), (2, 1, 1), (0, 0, 0), False),
}
CHANNELS = [(5, 16), (16, 32), (32, 64), (64, 64), (64, 128), (128, 128), (256, 256)]


def _random_sparse(rng, batch, shape, n, c):
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=min(n, cells), replace=False)
    rng.shuffle(lin)  # rows in arbitrary order, as a voxelizer delivers them
    b, r = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    idx = np.stack([b, z, y, x], 1).astype(np.int32)
    return idx, rng.standard_normal((len(lin), c)).astype(np.float32)


def _dense_reference(idx, feat, w5, bias, batch, shape, ks, st, pd, subm, go_fn):
    """fp64 dense conv on the CPU + autograd; returns (out_idx, out_feat, grad_in_rows, grad_w, go)."""
    i = torch.from_numpy(idx).long()
    x = torch.zeros(batch, feat.shape[1], *shape, dtype=torch.float64)
    x[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] = torch.from_numpy(feat).double()
    x.requires_grad_(True)
    w = torch.from_numpy(w5).double().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)  # [Cout,Cin,kd,kh,kw]
    b = None if bias is None else torch.from_numpy(bias).double()
    if subm:
        y = F.conv3d(x, w, b, 1, tuple(k // 2 for k in ks))
        o = i
    else:
        y = F.conv3d(x, w, b, st, pd)
        occ = torch.zeros(batch, 1, *shape, dtype=torch.float64)
        occ[i[:, 0], 0, i[:, 1], i[:, 2], i[:, 3]] = 1
        act = F.conv3d(occ, torch.ones(1, 1, *ks, dtype=torch.float64), None, st, pd) > 0
        o = torch.nonzero(act[:, 0])  # ascending (b, z, y, x)
    out = y[o[:, 0], :, o[:, 1], o[:, 2], o[:, 3]]
    go = go_fn(out.shape)
    out.backward(torch.from_numpy(go).double())
    gin = x.grad[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]]
    gw = w.grad.permute(0, 2, 3, 4, 1)  # back to [Cout,kd,kh,kw,Cin]
    return o.numpy().astype(np.int32), out.detach().numpy(), gin.numpy(), gw.numpy(), go, list(y.shape[2:])


def _close(name, got, want, rel=2e-5):
    scale = max(float(np.abs(want).max()), 1e-6)
    err = float(np.abs(got.astype(np.float64) - want).max())
    assert err <= rel * scale, "%s: max abs err %.3e vs scale %.3e (rel %.1e)" % (name, err, scale, rel)


@pytest.mark.parametrize("geom", list(GEOMS))
@pytest.mark.parametrize("cin,cout", CHANNELS)
def test_sparse_conv_equals_fp64_dense_conv3d(dev, geom, cin, cout):
    import efg_amd.spconv as spconv

    ks, st, pd, subm = GEOMS[geom]
    big = cin * cout >= 128 * 128
    batch, shape = 2, ((5, 12, 14) if big else (9, 20, 22))
    rng = np.random.default_rng(cin * 1000 + cout + len(geom))
    idx, feat = _random_sparse(rng, batch, shape, 500 if big else 1800, cin)
    bias = rng.standard_normal(cout).astype(np.float32) if (cin + cout) % 3 == 0 else None
    cls = spconv.SubMConv3d if subm else spconv.SparseConv3d
    kw = dict(indice_key="k") if subm else dict(stride=st)
    conv = cls(cin, cout, ks, padding=pd, bias=bias is not None, **kw).to(dev)
    w5 = (rng.standard_normal((cout, *ks, cin)) / np.sqrt(cin * ks[0] * ks[1] * ks[2])).astype(np.float32)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w5))
        if bias is not None:
            conv.bias.copy_(torch.from_numpy(bias))
    x = spconv.SparseConvTensor(torch.from_numpy(feat).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev),
                                list(shape), batch)
    y = conv(x)
    o_idx, o_feat, gin, gw, go, oshape = _dense_reference(
        idx, feat, w5, bias, batch, shape, ks, st, pd, subm,
        lambda s: np.random.default_rng(1).standard_normal(s).astype(np.float32))
    assert list(y.spatial_shape) == (list(shape) if subm else oshape)
    assert np.array_equal(y.indices.cpu().numpy(), o_idx), "active output sites / row order differ from conv3d(occupancy)"
    _close("forward", y.features.detach().cpu().numpy(), o_feat)
    y.features.backward(torch.from_numpy(go).to(dev))
    _close("dgrad", x.features.grad.cpu().numpy(), gin)
    _close("wgrad", conv.weight.grad.cpu().numpy(), gw, rel=5e-5)
    if bias is not None:
        _close("bias grad", conv.bias.grad.cpu().numpy(), go.astype(np.float64).sum(0), rel=5e-5)


def test_backbone_block_equals_dense(dev):
    """A res18 residual stage as the backbone wires it (strided main + shortcut convs sharing one geometry, SubM convs
    sharing an indice_key, BatchNorm1d over rows, residual add; sparse_net.py:120-165) against the same block built
    from dense fp64 conv3d with the activity mask applied after every layer."""
    import efg_amd.spconv as spconv
    from efg_amd.modeling.backbones.sparse_net import SparseBasicResBlock

    rng = np.random.default_rng(3)
    batch, shape, cin, cout = 2, (9, 18, 20), 32, 64
    idx, feat = _random_sparse(rng, batch, shape, 1500, cin)
    blk = SparseBasicResBlock(cin, cout, stride=2, norm="BN1d", activation=dict(type="ReLU", inplace=True),
                              indice_key="res2").to(dev)
    blk.train()
    x = spconv.SparseConvTensor(torch.from_numpy(feat).to(dev), torch.from_numpy(idx).to(dev), list(shape), batch)
    y = blk(x)
    # dense twin in fp64 on the CPU
    i = torch.from_numpy(idx).long()
    xd = torch.zeros(batch, cin, *shape, dtype=torch.float64)
    xd[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] = torch.from_numpy(feat).double()
    occ = torch.zeros(batch, 1, *shape, dtype=torch.float64)
    occ[i[:, 0], 0, i[:, 1], i[:, 2], i[:, 3]] = 1
    act = (F.conv3d(occ, torch.ones(1, 1, 3, 3, 3, dtype=torch.float64), None, 2, 1) > 0)
    sites = torch.nonzero(act[:, 0])

    def w_of(conv):
        return conv.weight.detach().double().cpu().permute(0, 4, 1, 2, 3)

    def bn_rows(rows, bn):  # batch statistics over the active rows, as BatchNorm1d over [M, C] in training mode
        m, v = rows.mean(0), rows.var(0, unbiased=False)
        return (rows - m) / torch.sqrt(v + bn.eps) * bn.weight.detach().double().cpu() + bn.bias.detach().double().cpu()

    def rows_of(d):
        return d[sites[:, 0], :, sites[:, 1], sites[:, 2], sites[:, 3]]

    def dense_of(rows, like):
        d = torch.zeros_like(like)
        d[sites[:, 0], :, sites[:, 1], sites[:, 2], sites[:, 3]] = rows
        return d

    conv_mods = [m for m in blk.conv._modules.values() if isinstance(m, spconv.SparseModule)]
    bns = [m for m in blk.conv._modules.values() if isinstance(m, torch.nn.BatchNorm1d)]
    h = F.conv3d(xd, w_of(conv_mods[0]), None, 2, 1)
    r = torch.relu(bn_rows(rows_of(h), bns[0]))
    h = F.conv3d(dense_of(r, h), w_of(conv_mods[1]), None, 1, 1)
    r = bn_rows(rows_of(h), bns[1])
    s = F.conv3d(xd, w_of(blk.shortcut[0]), None, 2, 1)
    rs = bn_rows(rows_of(s), blk.shortcut[1])
    want = torch.relu(r + rs).numpy()
    assert np.array_equal(y.indices.cpu().numpy(), sites.numpy().astype(np.int32))
    _close("residual block", y.features.detach().cpu().numpy(), want, rel=1e-4)


@pytest.mark.parametrize("stride", [1, 2])
def test_bottleneck_block_equals_dense(dev, stride):
    """The depth-50 block (1x1x1 SubM -> 3x3x3 -> 1x1x1 SubM + the basic block's shortcut; reference
    sparse_net.py:168-237) against the same block built from dense fp64 conv3d, forward and input gradient."""
    import efg_amd.spconv as spconv
    from efg_amd.modeling.backbones.sparse_net import SparseBottleneckBlock

    rng = np.random.default_rng(5 + stride)
    batch, shape, cin, mid = 2, (9, 18, 20), 64, 32
    cout = 64 if stride == 1 else 128
    idx, feat = _random_sparse(rng, batch, shape, 1500, cin)
    blk = SparseBottleneckBlock(cin, cout, mid, stride=stride, norm="BN1d", activation=dict(type="ReLU", inplace=True),
                                indice_key="res2").to(dev)
    blk.train()
    fx = torch.from_numpy(feat).to(dev).requires_grad_(True)
    x = spconv.SparseConvTensor(fx, torch.from_numpy(idx).to(dev), list(shape), batch)
    y = blk(x)
    i = torch.from_numpy(idx).long()
    xd = torch.zeros(batch, cin, *shape, dtype=torch.float64)
    xd[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] = torch.from_numpy(feat).double()
    xd.requires_grad_(True)
    occ = torch.zeros(batch, 1, *shape, dtype=torch.float64)
    occ[i[:, 0], 0, i[:, 1], i[:, 2], i[:, 3]] = 1
    act_in = occ[:, 0] > 0
    act_out = (F.conv3d(occ, torch.ones(1, 1, 3, 3, 3, dtype=torch.float64), None, 2, 1) > 0)[:, 0] if stride == 2 else act_in
    # rows of a submanifold layer keep the INPUT order; a strided layer's output sites are in canonical (sorted) order
    sites_in = i
    sites_out = i if stride == 1 else torch.nonzero(act_out)

    def w_of(conv):
        return conv.weight.detach().double().cpu().permute(0, 4, 1, 2, 3)

    def bn_rows(rows, bn):
        m, v = rows.mean(0), rows.var(0, unbiased=False)
        return (rows - m) / torch.sqrt(v + bn.eps) * bn.weight.detach().double().cpu() + bn.bias.detach().double().cpu()

    def rows_of(d, sites):
        return d[sites[:, 0], :, sites[:, 1], sites[:, 2], sites[:, 3]]

    def dense_of(rows, sites, spatial):
        out = torch.zeros(batch, rows.shape[1], *spatial, dtype=torch.float64)
        out[sites[:, 0], :, sites[:, 1], sites[:, 2], sites[:, 3]] = rows
        return out

    convs = [m for m in blk.conv._modules.values() if isinstance(m, spconv.SparseModule)]
    bns = [m for m in blk.conv._modules.values() if isinstance(m, torch.nn.BatchNorm1d)]
    h = F.conv3d(xd, w_of(convs[0]), None, 1, 0)                           # 1x1x1 on the input sites
    r = torch.relu(bn_rows(rows_of(h, sites_in), bns[0]))
    h = F.conv3d(dense_of(r, sites_in, shape), w_of(convs[1]), None, stride, 1)
    out_spatial = tuple(h.shape[2:])
    r = torch.relu(bn_rows(rows_of(h, sites_out), bns[1]))
    h = F.conv3d(dense_of(r, sites_out, out_spatial), w_of(convs[2]), None, 1, 0)
    r = bn_rows(rows_of(h, sites_out), bns[2])
    if blk.shortcut is not None:
        s = F.conv3d(xd, w_of(blk.shortcut[0]), None, stride, 1)
        rs = bn_rows(rows_of(s, sites_out), blk.shortcut[1])
    else:
        rs = rows_of(xd, sites_out)
    want = torch.relu(r + rs)
    assert np.array_equal(y.indices.cpu().numpy(), sites_out.numpy().astype(np.int32))
    _close("bottleneck block", y.features.detach().cpu().numpy(), want.detach().numpy(), rel=1e-4)
    go = torch.from_numpy(rng.standard_normal(tuple(want.shape)).astype(np.float32))
    y.features.backward(go.to(dev))
    want.backward(go.double())
    _close("bottleneck block input gradient", fx.grad.cpu().numpy(), rows_of(xd.grad, sites_in).numpy(), rel=2e-4)


def test_conv_bn_single_node_is_the_same_computation(dev, monkeypatch):
    """spconv.conv_bn_act runs the convolution and the fused BatchNorm (+ residual + ReLU) as ONE autograd node built from
    the two existing Functions (efg_amd/_fuse.py): same kernels, same order -> outputs, running statistics and every
    gradient identical bit for bit to the module-by-module form.  (The main + shortcut PAIR node sums the two input gradients
    inside its kernel, i.e. in another order: off here, held to rounding in tests/test_spconv_gpu.py.)"""
    monkeypatch.setenv("EFG_CONV_PAIR", "0")
    import efg_amd.spconv as spconv
    from efg_amd.modeling.backbones.sparse_net import SparseBasicResBlock
    from efg_amd.spconv import core

    rng = np.random.default_rng(11)
    batch, shape, cin, cout = 2, (9, 18, 20), 32, 64
    idx, feat = _random_sparse(rng, batch, shape, 1500, cin)
    go = None
    results = []
    for fused in (False, True):
        torch.manual_seed(0)
        blk = SparseBasicResBlock(cin, cout, stride=2, norm="BN1d", activation=dict(type="ReLU", inplace=True),
                                  indice_key="res2").to(dev)
        blk.train()
        saved = core._CONV_BN_FUSED
        core._CONV_BN_FUSED = fused
        try:
            fx = torch.from_numpy(feat).to(dev).requires_grad_(True)
            y = blk(spconv.SparseConvTensor(fx, torch.from_numpy(idx).to(dev), list(shape), batch))
            if go is None:
                go = torch.from_numpy(rng.standard_normal(tuple(y.features.shape)).astype(np.float32)).to(dev)
            y.features.backward(go)
        finally:
            core._CONV_BN_FUSED = saved
        results.append((y.features.detach().cpu().numpy(), fx.grad.cpu().numpy(),
                        {k: p.grad.cpu().numpy() for k, p in blk.named_parameters()},
                        {k: b.cpu().numpy() for k, b in blk.named_buffers()}))
    (y0, gx0, gp0, bf0), (y1, gx1, gp1, bf1) = results
    assert np.array_equal(y0, y1) and np.array_equal(gx0, gx1)
    assert all(np.array_equal(gp0[k], gp1[k]) for k in gp0) and set(gp0) == set(gp1)
    assert all(np.array_equal(bf0[k], bf1[k]) for k in bf0)
