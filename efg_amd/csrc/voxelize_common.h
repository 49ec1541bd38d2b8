// Shared pieces of the voxelizers (csrc/voxelize.hip: entry points + dynamic_voxelize; voxelize_bins.hip: the
// supercell-binned hard voxelizer; voxelize_hash.hip: the round 1-3 global-hash hard voxelizer kept for A/B runs and
// for grids too large to bin).
#pragma once
#include <string.h>

#include <algorithm>

#include "common.h"

namespace efg {

constexpr int kMaxBatch = 64;

struct VoxGeom {
  float vs[3];
  float rmin[3];
  int grid[3];  // x, y, z
};

struct SceneOffsets {
  long long off[kMaxBatch + 1];
};

// c = floor((p - min) / vs) per axis in IEEE fp32 (true division, no contraction): a point on a
// voxel boundary must land in the same voxel as on the CPU (voxelization_cpu.cpp:24).
__device__ __forceinline__ bool cell_of(float x, float y, float z, const VoxGeom& g, int& cx, int& cy, int& cz) {
  const float vx = floorf(__fdiv_rn(__fsub_rn(x, g.rmin[0]), g.vs[0]));
  const float vy = floorf(__fdiv_rn(__fsub_rn(y, g.rmin[1]), g.vs[1]));
  const float vz = floorf(__fdiv_rn(__fsub_rn(z, g.rmin[2]), g.vs[2]));
  // NaN fails every comparison -> outside
  const bool ok = (vx >= 0.0f) && (vx < (float)g.grid[0]) && (vy >= 0.0f) && (vy < (float)g.grid[1]) &&
                  (vz >= 0.0f) && (vz < (float)g.grid[2]);
  cx = (int)vx;
  cy = (int)vy;
  cz = (int)vz;
  return ok;
}

__device__ __forceinline__ bool point_cell(const float* __restrict__ p, const VoxGeom& g, int& cx, int& cy,
                                           int& cz) {
  return cell_of(p[0], p[1], p[2], g, cx, cy, cz);
}

// arguments of efg_hard_voxelize_f32 after validation (host side)
struct HardArgs {
  const float* points;
  SceneOffsets so;
  int64_t n_total, max_scene;
  int batch, f, max_points, max_voxels, coors_cols;
  VoxGeom g;
  unsigned long long vol;
  float* voxels;
  int32_t* coors;
  int32_t* npv;
  int32_t* voxel_num;
  float* mean;
  void* ws;
  size_t ws_bytes;
  hipStream_t stream;
};

size_t hash_workspace_bytes(int64_t n_total, int batch, int max_points, int max_voxels);
int hash_hard_voxelize(const HardArgs& a);
// 0 when the grid cannot be binned (too many supercells for the point count): the caller takes the hash path
size_t bins_workspace_bytes(int64_t n_total, int batch, int f, const VoxGeom& g);
int bins_hard_voxelize(const HardArgs& a);
void bins_set_debug_timeline(unsigned long long* buf);
size_t bins_debug_timeline_words();

}  // namespace efg
