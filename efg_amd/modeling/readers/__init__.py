from .voxel_reader import VoxelMeanFeatureExtractor  # noqa: F401
