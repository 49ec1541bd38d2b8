// oracle/_ref builder shim (OURS; test infrastructure, never shipped on the product path).
//
// Exposes the *reference's own* CPU voxelizers and its CPU point -> voxel grouping -- compiled from the sources
// where they lie under /root/reference (efg/operators/src/voxelize/voxelization_cpu.cpp:105-169,
// scatter_points_cpu.cpp:62-119) --
// through a plain C ABI so tests can call them with ctypes.  Only this shim lives in the
// repo; the reference translation unit is compiled in place by oracle/Makefile and the
// resulting shared object goes to oracle/_ref/ (git-ignored, travels with gpurun).
#include <torch/torch.h>
#include <cstring>
#include <vector>

namespace efg {
int hard_voxelize_cpu(const at::Tensor& points, at::Tensor& voxels, at::Tensor& coors,
                      at::Tensor& num_points_per_voxel, const std::vector<float> voxel_size,
                      const std::vector<float> coors_range, const int max_points,
                      const int max_voxels, const int NDim);
void dynamic_voxelize_cpu(const at::Tensor& points, at::Tensor& coors,
                          const std::vector<float> voxel_size,
                          const std::vector<float> coors_range, const int NDim);
// efg/operators/src/voxelize/scatter_points_cpu.cpp:62-119 (compiles, but the reference never binds it)
std::vector<at::Tensor> dynamic_point_to_voxel_cpu(const at::Tensor& points, const at::Tensor& voxel_mapping,
                                                   const std::vector<float> voxel_size,
                                                   const std::vector<float> coors_range);
}  // namespace efg

static std::vector<at::Tensor> g_dp2v;  // result of the last ref_dynamic_point_to_voxel_run (sizes are data dependent)

extern "C" {

// The reference's CPU point -> voxel grouping: every point of a voxel kept (no cap), voxels in first-occurrence order.
// coors i32 [n,3] (z,y,x), all inside the grid (the reference kernel does not skip -1 rows when it scatters).
// Returns voxel_num and writes max_points; fetch copies voxels [voxel_num, max_points, f], voxel_coors [voxel_num, 3],
// num_points_per_voxel [voxel_num].
int ref_dynamic_point_to_voxel_run(const float* points, long n, int f, const int* coors, const float* voxel_size,
                                   const float* coors_range, int* max_points) {
  at::Tensor pts = at::from_blob(const_cast<float*>(points), {n, f}, at::TensorOptions().dtype(at::kFloat));
  at::Tensor co = at::from_blob(const_cast<int*>(coors), {n, 3}, at::TensorOptions().dtype(at::kInt));
  std::vector<float> vs(voxel_size, voxel_size + 3), cr(coors_range, coors_range + 6);
  g_dp2v = efg::dynamic_point_to_voxel_cpu(pts, co, vs, cr);
  *max_points = (int)g_dp2v[0].size(1);
  return (int)g_dp2v[0].size(0);
}

void ref_dynamic_point_to_voxel_fetch(float* voxels, int* voxel_coors, int* num_points_per_voxel) {
  at::Tensor v = g_dp2v[0].contiguous(), c = g_dp2v[1].contiguous(), k = g_dp2v[2].contiguous();
  memcpy(voxels, v.data_ptr<float>(), sizeof(float) * v.numel());
  memcpy(voxel_coors, c.data_ptr<int>(), sizeof(int) * c.numel());
  memcpy(num_points_per_voxel, k.data_ptr<int>(), sizeof(int) * k.numel());
  g_dp2v.clear();
}

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
