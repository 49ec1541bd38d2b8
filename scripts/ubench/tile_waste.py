"""How many kernel offsets does a 16-row tile of the sparse-conv kernel execute, against what its rows need?
(union of the rows' active offsets per tile vs the mean per row) -- forward tables and dgrad (transposed) tables.
GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import efg_amd.spconv as spconv  # noqa: E402
from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene  # noqa: E402
from efg_amd.modeling.backbones import build_sparse_resnet_backbone  # noqa: E402
from efg_amd.operators import voxelize_batch  # noqa: E402

dev = torch.device("cuda:0")
pts = [torch.from_numpy(make_scene(2000 + i)[0]).to(dev) for i in range(2)]
vox = voxelize_batch(pts, VOXEL_SIZE, PC_RANGE, 5, 120000)
cfg = dict(depth=18, out_features=["res2", "res3", "res4"], num_groups=1, norm="BN1d",
           activation=dict(type="ReLU", inplace=True), width_per_group=64, res1_out_channels=64, stem_out_channels=32)
net = build_sparse_resnet_backbone(cfg, 5).to(dev)
net.dense_features = ["res3", "res4"]
import efg_amd.spconv.core as core  # noqa: E402

seen = []
_orig_init = core.Rulebook.__init__


def _spy(self, *a, **k):
    _orig_init(self, *a, **k)
    seen.append(self)


core.Rulebook.__init__ = _spy
x = spconv.SparseConvTensor(vox["voxel_mean"], vox["coordinates"], [41, 1504, 1504], 2)
out = net.stem(x)
for stage, name in net.stages_and_names:
    out = stage(out)


def stats(tab, rows):
    valid = (tab[:, :rows] >= 0)  # [kvol, rows]
    per_row = valid.sum(0).float().mean().item()
    pad = (-rows) % 16
    v = torch.nn.functional.pad(valid, (0, pad)).view(valid.shape[0], -1, 16)
    per_tile = v.any(2).sum(0).float().mean().item()
    return per_row, per_tile


for key, rb in enumerate(seen):
    f = stats(rb.nbr, rb.m_out)
    line = "%-28s %s m_in %6d m_out %6d kvol %2d | fwd: %.2f offsets/row, %.2f executed/tile (x%.2f)" % (
        str(key), "subm   " if rb.subm else "strided", rb.m_in, rb.m_out, rb.kvol, f[0], f[1], f[1] / max(f[0], 1e-9))
    if not rb.subm:
        d = stats(rb.rnbr, rb.m_in)
        line += " | dgrad: %.2f / row, %.2f / tile (x%.2f)" % (d[0], d[1], d[1] / max(d[0], 1e-9))
    print(line)
