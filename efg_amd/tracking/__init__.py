"""TrajectoryFormer (BASELINE configs[4]) training step on the MI355X path: counterpart of
playground/tracking.3d/waymo/trajectoryformer/trajectoryformer.centerpoint/{trajectoryformer,transformer,pointnet,
losses}.py and modules/{utils,blocks}.py.  The rotated IoU / NMS it leans on are efg_amd/operators/iou3d_nms.py."""
from .trajectoryformer import TrajectoryFormer  # noqa: F401
