"""Expose this package under the import names the reference playground uses.

`install()` registers sys.modules aliases so that, unchanged, playground code such as
`from efg.operators import Voxelization`, `from efg.modeling.operators import BoxAttnFunction`
(the stale path at $CQ/modules/box_attention.py:7), `from efg import _C` (efg_amd/_C.py),
`from efg.data.augmentations3d import _dict_select` (the stale path at $CP1/voxelnet.py:9),
`import spconv.pytorch as spconv`,
`from efg.modeling.backbones.fpn import build_resnet_fpn_backbone` and
`from efg.modeling.readers.voxel_reader import VoxelMeanFeatureExtractor` resolve to the MI355X
implementations.  Nothing is installed unless the caller asks for it, and an existing real `efg` /
`spconv` package is never overwritten."""
import sys
import types


def _dict_select(dict_, inds):
    """Index every array of a (nested) annotation dict with `inds` -- what $CP1/voxelnet.py:9 imports from the stale
    path `efg.data.augmentations3d` (the function lives in efg/data/utils/misc.py:1-10 in the reference tree)."""
    for key, value in dict_.items():
        if isinstance(value, dict):
            _dict_select(value, inds)
            continue
        try:
            dict_[key] = value[inds]
        except IndexError:
            dict_[key] = value[inds[len(value)]]


def install(force=False):
    from . import _C, data, modeling, operators, spconv
    from .modeling import backbones, common, readers
    from .modeling.backbones import fpn, sparse_net
    from .modeling.readers import voxel_reader
    from .operators import box_attention_func, iou3d_nms, ms_deform_attn, scatter_points, voxelize

    aug3d = types.ModuleType("efg.data.augmentations3d")
    aug3d._dict_select = _dict_select
    from .data import gpu_pipeline

    for name in ("RandomFlip3D", "GlobalRotation", "GlobalScaling", "GlobalTranslation", "FilterByRange",
                 "PointShuffle", "Voxelization"):
        if hasattr(gpu_pipeline, name):
            setattr(aug3d, name, getattr(gpu_pipeline, name))
    # the playground's stale operator package: ConQueR takes BoxAttnFunction from it, TrajectoryFormer
    # takes nms_gpu / boxes_iou3d_gpu ($TF/trajectoryformer.py:8), Mask2Former the ms_deform_attn submodule
    stale_ops = types.ModuleType("efg.modeling.operators")
    stale_ops.__path__ = []
    for src in (operators, iou3d_nms):
        for name in dir(src):
            if not name.startswith("_"):
                setattr(stale_ops, name, getattr(src, name))
    table = {
        "efg._C": _C,
        "efg.data": data,
        "efg.data.augmentations3d": aug3d,
        "efg.operators": operators,
        "efg.operators.voxelize": voxelize,
        "efg.operators.scatter_points": scatter_points,
        "efg.operators.box_attention_func": box_attention_func,
        "efg.operators.ms_deform_attn": ms_deform_attn,
        "efg.operators.iou3d_nms": iou3d_nms,
        "efg.modeling": modeling,
        "efg.modeling.operators": stale_ops,
        "efg.modeling.operators.ms_deform_attn": ms_deform_attn,
        "efg.modeling.operators.iou3d_nms": iou3d_nms,
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
