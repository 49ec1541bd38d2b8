"""MI355X-native Voxel-DETR / ConQueR training path.  Importing the package has no side effects;
process-level runtime settings live in efg_amd.engine.configure_hip_runtime()."""
