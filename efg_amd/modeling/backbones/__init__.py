from .fpn import FPN, LastLevelMaxPool, build_resnet_fpn_backbone  # noqa: F401
from .sparse_net import (SparseBasicBlock, SparseBasicResBlock, SparseBasicStem, SparseResNet,  # noqa: F401
                         SpMiddleResNetFHD, build_sparse_resnet_backbone)
