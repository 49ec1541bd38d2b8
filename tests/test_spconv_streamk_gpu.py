"""GPU: the stream-K launch of the tiled sparse convolution (csrc/spconv_tiles.hip) -- units of (row tile, n-slice) cut
into equal shares of (unit, offset) items, shared units summed by the workgroup holding the last share.
Each variant runs in a child process (the library reads its switches once): stream-K off / on / on again / on with
every late share recomputed by its owner (EFG_TILE_SK_POLLS=0).  on == off to fp32 summation-order tolerance, on == on
bit for bit (the partition and the order of the shares are fixed), the recompute path == off to the same tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import efg_amd.spconv as spconv
from efg_amd.spconv import core
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
# clustered sites (a few planes + blobs): neighbour masks of every weight, like a BEV scene
shape, batch = (12, 160, 160), 2
pts = []
for b in range(batch):
    z = rng.integers(0, shape[0], 60000); y = rng.integers(0, shape[1], 60000); x = rng.integers(0, shape[2], 60000)
    keep = (z < 3) | ((x + y) %% 7 == 0) | (rng.random(60000) < 0.08)
    pts.append(np.stack([np.full(keep.sum(), b), z[keep], y[keep], x[keep]], 1))
idx = np.unique(np.concatenate(pts), axis=0).astype(np.int32)
out = {}
for cin, cout, strided in ((64, 64, False), (128, 128, False), (256, 256, False), (128, 256, True)):
    torch.manual_seed(cin + cout)
    feat = torch.randn(idx.shape[0], cin, device=dev)
    x = spconv.SparseConvTensor(feat, torch.from_numpy(idx).to(dev), list(shape), batch)
    conv = (spconv.SparseConv3d(cin, cout, 3, 2, padding=1, bias=True) if strided else
            spconv.SubMConv3d(cin, cout, 3, padding=1, bias=True, indice_key="k%%d" %% cin)).to(dev)
    xin = x.replace_feature(feat.clone().requires_grad_(True))
    y = conv(xin)
    go = torch.randn(y.features.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    y.features.backward(go)
    key = "%%d_%%d_%%d" %% (cin, cout, strided)
    out["y_" + key] = y.features.detach().cpu().numpy()
    out["dx_" + key] = xin.features.grad.cpu().numpy()
    out["rows_" + key] = np.array([idx.shape[0], y.features.shape[0]])
torch.cuda.synchronize()
np.savez(sys.argv[1], **out)
print("CHILD_OK")
"""


def _run(tmp_path, name, env):
    path = str(tmp_path / (name + ".npz"))
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **env)
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}, path], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=e)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (name, r.stdout[-1000:], r.stderr[-3000:])
    return dict(np.load(path))


def test_streamk_matches_plain_launch_and_is_reproducible(tmp_path):
    off = _run(tmp_path, "off", {"EFG_TILE_STREAMK": "0"})
    on = _run(tmp_path, "on", {"EFG_TILE_STREAMK": "2"})        # 2: every eligible shape, also the strided ones
    again = _run(tmp_path, "again", {"EFG_TILE_STREAMK": "2"})
    redo = _run(tmp_path, "redo", {"EFG_TILE_STREAMK": "2", "EFG_TILE_SK_POLLS": "0"})
    assert int(off["rows_64_64_0"][0]) > 20000                  # enough items for every workgroup of the launch
    for k in off:
        if k.startswith("rows_"):
            continue
        scale = float(np.abs(off[k]).max())
        np.testing.assert_allclose(on[k], off[k], rtol=2e-5, atol=2e-5 * scale, err_msg=k)
        assert np.array_equal(on[k], again[k]), "%s differs between two stream-K runs" % k
        np.testing.assert_allclose(redo[k], off[k], rtol=2e-5, atol=2e-5 * scale, err_msg=k + " (recompute path)")


def test_split_precision_arm_of_the_tile_kernel(tmp_path):
    """EFG_GEMM_ARM=bf16x3 (the bench's A/B arm): forward and data gradient of the 64-channel-wide split-K shapes as three bf16
    MFMA products of split operands (MODE & 4 of conv_tile_kernel), plain and stream-K launch, against the exact fp32 kernel."""
    off = _run(tmp_path, "fp32", {"EFG_TILE_STREAMK": "0"})
    arm = _run(tmp_path, "arm", {"EFG_TILE_STREAMK": "0", "EFG_GEMM_ARM": "bf16x3"})
    arm_sk = _run(tmp_path, "arm_sk", {"EFG_TILE_STREAMK": "2", "EFG_GEMM_ARM": "bf16x3"})
    again = _run(tmp_path, "arm_sk2", {"EFG_TILE_STREAMK": "2", "EFG_GEMM_ARM": "bf16x3"})
    changed = 0
    for k in off:
        if k.startswith("rows_"):
            continue
        scale = float(np.abs(off[k]).max())
        for name, got in (("arm", arm), ("arm + stream-K", arm_sk)):
            err = float(np.abs(got[k] - off[k]).max()) / scale
            assert err < 3e-5, (k, name, err)       # 16 significand bits per operand; a wrong lane order would be O(1)
            changed += err > 0
        assert np.array_equal(arm_sk[k], again[k]), "%s differs between two runs" % k
    assert changed >= 8                              # the arm really ran (fp32 against itself would be exactly 0)
