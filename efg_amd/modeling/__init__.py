"""`efg.modeling` subset on the hot path (sparse backbones, FPN neck, voxel reader)."""
