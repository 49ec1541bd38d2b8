// Point-cloud augmentation + range filter on the GPU (the step BEFORE the path).
//
// The reference augments every cloud on DataLoader workers with numpy (efg/data/augmentations/extend_3d.py:
// RandomFlip3D :120-162, GlobalRotation :165-199, GlobalScaling :202-218, GlobalTranslation :221-236,
// FilterByRange :286-315, PointShuffle :108-118), one pass over the cloud per processor.  Here the per-point
// transforms are an op list applied in registers, in the reference's order and fp32 arithmetic, fused with the
// range filter and an ORDER-PRESERVING compaction (points[keep]); the shuffle is a row gather by a permutation.
#include "common.h"

namespace efg {
namespace {

constexpr int kMaxOps = 8;
constexpr int kTile = 1024;  // points per workgroup (256 threads x 4 consecutive points)

struct OpList {
  int n;
  int kind[kMaxOps];
  float a[kMaxOps], b[kMaxOps], c[kMaxOps];
};

struct Range {
  int on;
  float lo[3], hi[3];
};

__device__ __forceinline__ void apply_ops(const OpList& ops, float& x, float& y, float& z) {
  for (int i = 0; i < ops.n; ++i) {
    switch (ops.kind[i]) {
      case EFG_PT_NEG_Y: y = -y; break;
      case EFG_PT_NEG_X: x = -x; break;
      case EFG_PT_ROT_Z: {  // [x, y] @ [[c, s], [-s, c]]  (rotate_points_along_z, box_ops.py:527-532)
        const float nx = x * ops.a[i] + y * (-ops.b[i]);
        const float ny = x * ops.b[i] + y * ops.a[i];
        x = nx;
        y = ny;
        break;
      }
      case EFG_PT_SCALE:
        x *= ops.a[i];
        y *= ops.a[i];
        z *= ops.a[i];
        break;
      case EFG_PT_TRANSLATE:
        x += ops.a[i];
        y += ops.b[i];
        z += ops.c[i];
        break;
    }
  }
}

__device__ __forceinline__ bool in_range(const Range& r, float x, float y, float z) {  // box_ops.py:538-548
  return !r.on || (x >= r.lo[0] && x <= r.hi[0] && y >= r.lo[1] && y <= r.hi[1] && z >= r.lo[2] && z <= r.hi[2]);
}

__global__ void __launch_bounds__(256)
pts_count_kernel(const float* __restrict__ in, long long n, int f, OpList ops, Range rg, int* __restrict__ counts) {
  __shared__ int sm[17];
  const long long base = (long long)blockIdx.x * kTile + threadIdx.x * 4;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = base + j;
    if (i < n) {
      float x = in[i * f], y = in[i * f + 1], z = in[i * f + 2];
      apply_ops(ops, x, y, z);
      cnt += in_range(rg, x, y, z) ? 1 : 0;
    }
  }
  int tot;
  block_exclusive_scan(cnt, sm, &tot);
  if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

// exclusive scan of the per-tile counts (one workgroup, any length) + the grand total
__global__ void __launch_bounds__(1024) pts_scan_kernel(int* __restrict__ counts, int ntiles, int* __restrict__ total) {
  __shared__ int sm[17];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < ntiles ? counts[i] : 0;
    int tot;
    const int pre = block_exclusive_scan(v, sm, &tot);
    const int carry = carry_s;
    if (i < ntiles) counts[i] = carry + pre;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ void __launch_bounds__(256)
pts_write_kernel(const float* __restrict__ in, long long n, int f, OpList ops, Range rg,
                 const int* __restrict__ offsets, float* __restrict__ out) {
  __shared__ int sm[17];
  const long long base = (long long)blockIdx.x * kTile + threadIdx.x * 4;
  float xs[4], ys[4], zs[4];
  bool keep[4];
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = base + j;
    keep[j] = false;
    if (i < n) {
      xs[j] = in[i * f];
      ys[j] = in[i * f + 1];
      zs[j] = in[i * f + 2];
      apply_ops(ops, xs[j], ys[j], zs[j]);
      keep[j] = in_range(rg, xs[j], ys[j], zs[j]);
      cnt += keep[j] ? 1 : 0;
    }
  }
  int tot;
  // without a filter every point is kept: tile b starts at b * kTile
  long long pos = (offsets ? (long long)offsets[blockIdx.x] : (long long)blockIdx.x * kTile) +
                  block_exclusive_scan(cnt, sm, &tot);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (keep[j]) {
      const long long i = base + j;
      float* o = out + pos * f;
      o[0] = xs[j];
      o[1] = ys[j];
      o[2] = zs[j];
      for (int k = 3; k < f; ++k) o[k] = in[i * f + k];
      ++pos;
    }
  }
}

__global__ void __launch_bounds__(256)
pts_gather_kernel(const float* __restrict__ in, const long long* __restrict__ index, long long m, int f,
                  float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * f) return;
  const long long row = e / f;
  const int k = (int)(e % f);
  out[e] = in[index[row] * f + k];
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" size_t efg_points_transform_filter_workspace_bytes(int64_t n) {
  return align_up(sizeof(int) * (size_t)ceil_div(n > 0 ? n : 1, kTile), 256);
}

extern "C" int efg_points_transform_filter_f32(const float* points, int64_t n, int f, const efg_point_op* ops_host,
                                               int n_ops, const float* range_host, float* out, int32_t* count,
                                               void* ws, size_t ws_bytes, void* stream) {
  EFG_CHECK_ARG(n >= 0 && f >= 3, "points_transform_filter: need n >= 0 and at least 3 columns (got %d)", f);
  EFG_CHECK_ARG(n_ops >= 0 && n_ops <= kMaxOps, "points_transform_filter: at most %d ops (got %d)", kMaxOps, n_ops);
  EFG_CHECK_ARG(count, "points_transform_filter: count is null");
  EFG_CHECK_ARG(n < (1ll << 31), "points_transform_filter: too many points");
  hipStream_t st = (hipStream_t)stream;
  OpList ops{};
  ops.n = n_ops;
  for (int i = 0; i < n_ops; ++i) {
    EFG_CHECK_ARG(ops_host[i].kind >= EFG_PT_NEG_Y && ops_host[i].kind <= EFG_PT_TRANSLATE,
                  "points_transform_filter: unknown op kind %d", ops_host[i].kind);
    ops.kind[i] = ops_host[i].kind;
    ops.a[i] = ops_host[i].a;
    ops.b[i] = ops_host[i].b;
    ops.c[i] = ops_host[i].c;
  }
  Range rg{};
  rg.on = range_host != nullptr;
  for (int k = 0; k < 3 && range_host; ++k) {
    rg.lo[k] = range_host[k];
    rg.hi[k] = range_host[3 + k];
  }
  if (n == 0) {
    EFG_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t), st));
    return EFG_OK;
  }
  EFG_CHECK_ARG(points && out, "points_transform_filter: null pointer");
  const int ntiles = (int)ceil_div(n, kTile);
  if (!rg.on) {  // nothing to compact: one pass
    hipLaunchKernelGGL(pts_write_kernel, dim3(ntiles), dim3(256), 0, st, points, (long long)n, f, ops, rg,
                       (const int*)nullptr, out);
    EFG_LAUNCH_CHECK();
    const int32_t nn = (int32_t)n;
    EFG_HIP_TRY(hipMemcpyAsync(count, &nn, sizeof(int32_t), hipMemcpyHostToDevice, st));
    return EFG_OK;
  }
  EFG_CHECK_ARG(ws && ws_bytes >= efg_points_transform_filter_workspace_bytes(n),
                "points_transform_filter: workspace too small");
  int* counts = static_cast<int*>(ws);
  hipLaunchKernelGGL(pts_count_kernel, dim3(ntiles), dim3(256), 0, st, points, (long long)n, f, ops, rg, counts);
  hipLaunchKernelGGL(pts_scan_kernel, dim3(1), dim3(1024), 0, st, counts, ntiles, count);
  hipLaunchKernelGGL(pts_write_kernel, dim3(ntiles), dim3(256), 0, st, points, (long long)n, f, ops, rg,
                     (const int*)counts, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_points_gather_f32(const float* points, const int64_t* index, int64_t m, int f, float* out,
                                     void* stream) {
  EFG_CHECK_ARG(m >= 0 && f >= 1, "points_gather: bad sizes");
  if (m == 0) return EFG_OK;
  EFG_CHECK_ARG(points && index && out, "points_gather: null pointer");
  hipLaunchKernelGGL(pts_gather_kernel, dim3((unsigned)ceil_div(m * f, 256)), dim3(256), 0, (hipStream_t)stream, points,
                     (const long long*)index, (long long)m, f, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
