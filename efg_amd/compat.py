"""Expose this package under the import names the reference playground uses.

`install()` registers sys.modules aliases so that, unchanged, playground code such as
`from efg.operators import Voxelization`, `from efg.modeling.operators import BoxAttnFunction`
(the stale path at $CQ/modules/box_attention.py:7), `import spconv.pytorch as spconv`,
`from efg.modeling.backbones.fpn import build_resnet_fpn_backbone` and
`from efg.modeling.readers.voxel_reader import VoxelMeanFeatureExtractor` resolve to the MI355X
implementations.  Nothing is installed unless the caller asks for it, and an existing real `efg` /
`spconv` package is never overwritten."""
import sys
import types


def install(force=False):
    from . import modeling, operators, spconv
    from .modeling import backbones, common, readers
    from .modeling.backbones import fpn, sparse_net
    from .modeling.readers import voxel_reader
    from .operators import box_attention_func, iou3d_nms, ms_deform_attn, scatter_points, voxelize

    table = {
        "efg.operators": operators,
        "efg.operators.voxelize": voxelize,
        "efg.operators.scatter_points": scatter_points,
        "efg.operators.box_attention_func": box_attention_func,
        "efg.operators.ms_deform_attn": ms_deform_attn,
        "efg.operators.iou3d_nms": iou3d_nms,
        "efg.modeling": modeling,
        "efg.modeling.operators": operators,
        "efg.modeling.common": common,
        "efg.modeling.backbones": backbones,
        "efg.modeling.backbones.fpn": fpn,
        "efg.modeling.backbones.sparse_net": sparse_net,
        "efg.modeling.readers": readers,
        "efg.modeling.readers.voxel_reader": voxel_reader,
        "spconv": spconv,
        "spconv.pytorch": spconv,
    }
    if "efg" not in sys.modules or force:
        root = types.ModuleType("efg")
        root.__path__ = []
        root.__version__ = "1.0-mi355x"
        sys.modules["efg"] = root
    for name, mod in table.items():
        if name in sys.modules and not force and sys.modules[name] is not mod:
            continue
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, mod)
    return table
