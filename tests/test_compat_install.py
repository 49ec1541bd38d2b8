"""`efg_amd.compat.install()`: the import names the reference playground uses resolve to this package -- including
`efg._C` with the pybind signatures of efg/operators/src/vision.cpp:70-122 (in-scope functions callable, the others
importable and raising lazily), the stale `efg.modeling.operators` / `efg.data.augmentations3d` paths
 and `spconv.pytorch`.  Runs in a subprocess so that the aliases do not leak into other tests."""
import subprocess
import sys

import pytest

from conftest import ROOT

IMPORTS = r"""
import sys
sys.path.insert(0, %r)
import efg_amd.compat as compat
compat.install()
from efg.operators import Voxelization, voxelization, DynamicScatter, dynamic_scatter, BoxAttnFunction
from efg.modeling.operators import BoxAttnFunction as B2, nms_gpu, boxes_iou3d_gpu      # stale playground path
from efg.operators.ms_deform_attn import MSDeformAttn, MSDeformAttnFunction
import spconv.pytorch as spconv
from spconv.pytorch import SparseConv3d, SubMConv3d
from efg.modeling.backbones.fpn import build_resnet_fpn_backbone
from efg.modeling.backbones.sparse_net import SparseResNet, SpMiddleResNetFHD
from efg.modeling.readers.voxel_reader import VoxelMeanFeatureExtractor
from efg.data.augmentations3d import _dict_select                                        # $CP1/voxelnet.py:9
from efg import _C
from efg._C import dynamic_point_to_voxel_backward, dynamic_point_to_voxel_forward       # scatter_points.py:5
import numpy as np
d = {"a": np.arange(5), "n": {"b": np.arange(10, 15)}}
_dict_select(d, np.array([0, 2]))
assert d["a"].tolist() == [0, 2] and d["n"]["b"].tolist() == [10, 12]
for name in ("hard_voxelize", "dynamic_voxelize", "box_attn_forward", "box_attn_backward", "ms_deform_attn_forward",
             "ms_deform_attn_backward", "boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu"):
    assert callable(getattr(_C, name)), name
f = _C.deform_conv_forward          # importable ...
try:
    f()
    raise SystemExit("out-of-scope binding did not raise")
except NotImplementedError:
    pass                            # ... and raising only when called
try:
    _C.no_such_function
    raise SystemExit("unknown attribute resolved")
except AttributeError:
    pass
assert B2 is BoxAttnFunction and spconv.SparseConvTensor is not None
print("COMPAT_OK")
"""

GPU_CALLS = IMPORTS + r"""
import torch
dev = torch.device("cuda:0")
pts = torch.rand(5000, 5, device=dev) * torch.tensor([20., 20., 4., 1., 1.], device=dev) - torch.tensor([10., 10., 2., 0., 0.], device=dev)
voxels = torch.zeros(20000, 5, 5, device=dev); coors = torch.zeros(20000, 3, dtype=torch.int32, device=dev)
npv = torch.zeros(20000, dtype=torch.int32, device=dev)
n = _C.hard_voxelize(pts, voxels, coors, npv, [0.5, 0.5, 0.5], [-10., -10., -2., 10., 10., 2.], 5, 20000, 3)
assert isinstance(n, int) and 0 < n < 20000 and int(npv[:n].min()) >= 1 and int(npv[n:].sum()) == 0
dc = torch.zeros(5000, 3, dtype=torch.int32, device=dev)
_C.dynamic_voxelize(pts, dc, [0.5, 0.5, 0.5], [-10., -10., -2., 10., 10., 2.], 3)
vf, vc, p2v, cnt = _C.dynamic_point_to_voxel_forward(pts.contiguous(), dc, "mean")
assert vf.shape[0] == vc.shape[0] == n                                        # same occupied voxels either way
boxes = torch.rand(64, 7, device=dev) * torch.tensor([10., 10., 1., 3., 2., 1.5, 3.], device=dev) + torch.tensor([0., 0., 0., 1., 1., 1., 0.], device=dev)
iou = torch.zeros(64, 64, device=dev); _C.boxes_iou_bev_gpu(boxes, boxes, iou)
assert torch.allclose(iou.diag(), torch.ones(64, device=dev), atol=1e-4)
keep = torch.LongTensor(64); k = _C.nms_gpu(boxes, keep, 0.1)
assert 0 < k <= 64 and int(keep[0]) == 0
conv = SubMConv3d(5, 8, 3, padding=1, bias=False, indice_key="a").to(dev)
x = spconv.SparseConvTensor(vf, torch.nn.functional.pad(vc, (1, 0)), [8, 40, 40], 1)
assert conv(x).features.shape == (n, 8)
print("COMPAT_GPU_OK")
"""


def _run(code):
    return subprocess.run([sys.executable, "-c", code % ROOT], capture_output=True, text=True, timeout=600)


def test_install_imports_cpu():
    r = _run(IMPORTS)
    assert "COMPAT_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_install_and_call_gpu(dev):
    r = _run(GPU_CALLS)
    assert "COMPAT_GPU_OK" in r.stdout, r.stdout + r.stderr
