"""CenterPoint (BASELINE configs[0] and [3]) on the MI355X path: reader -> SpMiddleResNetFHD (HIP sparse convs) ->
RPN -> CenterHead, counterpart of playground/detection.3d/waymo/center_point/centerpoint.waymo.voxelnet.*/
{voxelnet,center_head,centernet_loss,center_utils}.py."""
from .center_head import CenterHead, FastFocalLoss, RegLoss, SepHead  # noqa: F401
from .voxelnet import VoxelNet  # noqa: F401
