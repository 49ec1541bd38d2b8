"""Voxel-DETR / ConQueR 3-D detector: the caller of the hot path (playground/detection.3d/waymo/conquer)."""
from .box_attention import Box3dAttention  # noqa: F401
from .transformer import Transformer, TransformerDecoder  # noqa: F401
from .voxel_detr import VoxelDETR  # noqa: F401
