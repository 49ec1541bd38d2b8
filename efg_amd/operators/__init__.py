"""`efg.operators` surface on MI355X (efg/operators/__init__.py:1-5)."""
from .box_attention_func import BoxAttnFunction  # noqa: F401
from .scatter_points import DynamicScatter, dynamic_scatter  # noqa: F401
from .voxelize import Voxelization, voxelization, voxelize_batch  # noqa: F401

__all__ = ["Voxelization", "voxelization", "dynamic_scatter", "DynamicScatter", "BoxAttnFunction", "voxelize_batch"]
