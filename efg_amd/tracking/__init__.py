"""TrajectoryFormer (BASELINE configs[4]) on the MI355X path -- training step and online tracker: counterpart of
playground/tracking.3d/waymo/trajectoryformer/trajectoryformer.centerpoint/{trajectoryformer,transformer,pointnet,
losses}.py and modules/{utils,blocks}.py.  The rotated IoU / NMS it leans on are efg_amd/operators/iou3d_nms.py."""
from .trajectoryformer import TrajectoryFormer  # noqa: F401
