"""efg/modeling/readers/voxel_reader.py:8-19."""
from torch import nn


class VoxelMeanFeatureExtractor(nn.Module):
    """Mean of the (<= max_points, zero padded) points of each voxel, divided by the CAPPED count.

    On the fused path the voxelizer already produced this (`voxel_mean`, csrc/voxelize.hip K5); the
    module form is kept for the reference call signature `reader(voxels, num_points, coords)`.
    """

    def __init__(self, num_input_features, norm="BN1d"):
        super().__init__()
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        points_mean = features[:, :, : self.num_input_features].sum(dim=1, keepdim=False) / num_voxels.type_as(
            features).view(-1, 1)
        return points_mean.contiguous()
