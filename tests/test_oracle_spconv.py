"""CPU: the oracle's sparse convolution against dense torch.nn.functional.conv3d (the contract of
SURVEY.md B.6 / §8c: spconv itself is absent, so parity for this row is UNPINNED by the reference;
the dense-equivalence below is what both our oracle and the HIP path are held to)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def random_sparse(rng, batch, shape, m, c):
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=m, replace=False)
    rng.shuffle(lin)
    x = lin % shape[2]
    y = (lin // shape[2]) % shape[1]
    z = (lin // (shape[2] * shape[1])) % shape[0]
    b = lin // (shape[2] * shape[1] * shape[0])
    idx = np.stack([b, z, y, x], 1).astype(np.int32)
    feat = rng.standard_normal((m, c)).astype(np.float32)
    return idx, feat


def dense_of(idx, feat, batch, shape):
    d = torch.zeros(batch, feat.shape[1], *shape, dtype=torch.float64)
    d[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.from_numpy(feat).double()
    return d


def w_dense(w, ks):
    cout, kvol, cin = w.shape
    return torch.from_numpy(w).double().view(cout, ks[0], ks[1], ks[2], cin).permute(0, 4, 1, 2, 3).contiguous()


CONVS = [  # (ksize, stride, pad) as used in sparse_net.py:86,136,277,500-523
    ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((3, 1, 1), (2, 1, 1), (1, 0, 0)),
    ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
    ((3, 1, 1), (2, 1, 1), (0, 0, 0)),
]


@pytest.mark.parametrize("ks,st,pd", CONVS)
def test_regular_conv_matches_dense(oracle_mod, ks, st, pd):
    rng = np.random.default_rng(0)
    batch, shape, cin, cout = 2, (9, 12, 10), 5, 7
    idx, feat = random_sparse(rng, batch, shape, 150, cin)
    w = rng.standard_normal((cout, ks[0] * ks[1] * ks[2], cin)).astype(np.float32)
    out_idx, oshape = oracle_mod.spconv_out_indices(idx, batch, shape, ks, st, pd)
    nbr = oracle_mod.spconv_rulebook(idx, out_idx, batch, shape, ks, st, pd)
    out = oracle_mod.spconv_forward(feat, w, None, nbr)
    dn = dense_of(idx, feat, batch, shape)
    ref = F.conv3d(dn, w_dense(w, ks), stride=st, padding=pd)
    occ = F.conv3d((dn.abs().sum(1, keepdim=True) > 0).double(), torch.ones(1, 1, *ks, dtype=torch.float64), stride=st,
                   padding=pd)
    assert list(ref.shape[2:]) == oshape
    act = torch.nonzero(occ[:, 0] > 0).numpy().astype(np.int32)  # sorted (b,z,y,x) = canonical order
    assert np.array_equal(act, out_idx)
    ref_rows = ref[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]].numpy()
    np.testing.assert_allclose(out, ref_rows, rtol=1e-5, atol=1e-5)
    # dense() of the oracle
    d = oracle_mod.sparse_to_dense(out, out_idx, batch, oshape)
    mask = (occ[:, 0] > 0).numpy()
    np.testing.assert_allclose(d * mask[:, None], (ref * torch.from_numpy(mask[:, None])).numpy(), rtol=1e-5,
                               atol=1e-5)


def test_subm_conv_and_grads_match_dense(oracle_mod):
    rng = np.random.default_rng(1)
    batch, shape, cin, cout = 2, (7, 9, 8), 6, 4
    ks, st, pd = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    idx, feat = random_sparse(rng, batch, shape, 200, cin)
    w = rng.standard_normal((cout, 27, cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    nbr = oracle_mod.spconv_rulebook(idx, idx, batch, shape, ks, st, pd)
    out = oracle_mod.spconv_forward(feat, w, b, nbr)
    x = torch.from_numpy(feat).double().requires_grad_(True)
    wd = torch.from_numpy(w).double().requires_grad_(True)
    dn = torch.zeros(batch, cin, *shape, dtype=torch.float64)
    dn[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = x
    ref = F.conv3d(dn, wd.view(cout, 3, 3, 3, cin).permute(0, 4, 1, 2, 3), torch.from_numpy(b).double(), padding=1)
    rows = ref[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    np.testing.assert_allclose(out, rows.detach().numpy(), rtol=1e-5, atol=1e-5)
    go = rng.standard_normal(out.shape).astype(np.float32)
    rows.backward(torch.from_numpy(go).double())
    gi = oracle_mod.spconv_dgrad(go, w, nbr, feat.shape[0])
    gw = oracle_mod.spconv_wgrad(feat, go, nbr)
    np.testing.assert_allclose(gi, x.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gw, wd.grad.numpy(), rtol=1e-5, atol=1e-5)
