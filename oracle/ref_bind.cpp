// oracle/_ref builder shim (OURS; test infrastructure, never shipped on the product path).
//
// Exposes the *reference's own* CPU voxelizers -- compiled from the sources where they
// lie under /root/reference (efg/operators/src/voxelize/voxelization_cpu.cpp:105-169) --
// through a plain C ABI so tests can call them with ctypes.  Only this shim lives in the
// repo; the reference translation unit is compiled in place by oracle/Makefile and the
// resulting shared object goes to oracle/_ref/ (git-ignored, travels with gpurun).
#include <torch/torch.h>
#include <vector>

namespace efg {
int hard_voxelize_cpu(const at::Tensor& points, at::Tensor& voxels, at::Tensor& coors,
                      at::Tensor& num_points_per_voxel, const std::vector<float> voxel_size,
                      const std::vector<float> coors_range, const int max_points,
                      const int max_voxels, const int NDim);
void dynamic_voxelize_cpu(const at::Tensor& points, at::Tensor& coors,
                          const std::vector<float> voxel_size,
                          const std::vector<float> coors_range, const int NDim);
}  // namespace efg

extern "C" {

int ref_hard_voxelize_cpu(const float* points, long n, int f, const float* voxel_size,
                          const float* coors_range, int max_points, int max_voxels,
                          float* voxels, int* coors, int* num_points_per_voxel) {
  auto fopt = at::TensorOptions().dtype(at::kFloat);
  auto iopt = at::TensorOptions().dtype(at::kInt);
  at::Tensor pts = at::from_blob(const_cast<float*>(points), {n, f}, fopt);
  at::Tensor vox = at::from_blob(voxels, {max_voxels, max_points, f}, fopt);
  at::Tensor co = at::from_blob(coors, {max_voxels, 3}, iopt);
  at::Tensor npv = at::from_blob(num_points_per_voxel, {max_voxels}, iopt);
  std::vector<float> vs(voxel_size, voxel_size + 3), cr(coors_range, coors_range + 6);
  return efg::hard_voxelize_cpu(pts, vox, co, npv, vs, cr, max_points, max_voxels, 3);
}

void ref_dynamic_voxelize_cpu(const float* points, long n, int f, const float* voxel_size,
                              const float* coors_range, int* coors) {
  auto fopt = at::TensorOptions().dtype(at::kFloat);
  auto iopt = at::TensorOptions().dtype(at::kInt);
  at::Tensor pts = at::from_blob(const_cast<float*>(points), {n, f}, fopt);
  at::Tensor co = at::from_blob(coors, {n, 3}, iopt);
  std::vector<float> vs(voxel_size, voxel_size + 3), cr(coors_range, coors_range + 6);
  efg::dynamic_voxelize_cpu(pts, co, vs, cr, 3);
}

}  // extern "C"
