// Points inside vertical cylinders, per cylinder in cloud order.
//
// TrajectoryFormer crops, for every hypothesis box of a scene, the current-sweep points whose BEV distance to the box
// centre is at most 1.2 x the half diagonal ($TF/modules/utils.py:361-431: a [boxes x points] distance matrix and a
// Python loop over boxes).  With 300-500 boxes and 180 000 points per scene that matrix is 220 MB per scene and its
// `norm`, compare and `nonzero` are 3.5 ms of a 59 ms training step.  Here a workgroup owns 16 cylinders and walks
// the scene's points once, 256 at a time: one coalesced point load serves 16 tests, the per-cylinder order-preserving
// positions come from wave ballots + a 4-wave prefix through LDS.  The same kernel runs twice: counts only (the host
// needs them anyway: it sizes the result and draws the reference's sub-sampling of crowded boxes), then the fill.
#include "common.h"

namespace efg {
namespace {

constexpr int kCyl = 16;  // cylinders per workgroup

__global__ void __launch_bounds__(256) cylinder_select_kernel(const float* __restrict__ points, int f, int time_col, float max_time,
                                                              const long long* __restrict__ range,  // [R][2] point rows
                                                              const float* __restrict__ xyr,         // [R][3]
                                                              long long n_cyl, const long long* __restrict__ starts,
                                                              int* __restrict__ counts, int* __restrict__ index,
                                                              int n_chunks, int* __restrict__ chunk_counts) {
  __shared__ float cx[kCyl], cy[kCyl], cr[kCyl];
  __shared__ int wave_cnt[4][kCyl];
  __shared__ long long chunk_base[kCyl];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long c0 = (long long)blockIdx.x * kCyl;
  const int nc = (int)min((long long)kCyl, n_cyl - c0);
  if (threadIdx.x < kCyl) {
    const bool ok = (int)threadIdx.x < nc;
    cx[threadIdx.x] = ok ? xyr[(c0 + threadIdx.x) * 3 + 0] : 0.f;
    cy[threadIdx.x] = ok ? xyr[(c0 + threadIdx.x) * 3 + 1] : 0.f;
    cr[threadIdx.x] = ok ? xyr[(c0 + threadIdx.x) * 3 + 2] : -1.f;  // nothing is inside a padding cylinder
  }
  // the cylinders of a workgroup belong to one scene (the host pads each scene's list to a multiple of 16)
  // blockIdx.y = which slice of the scene's points: 16 cylinders x one slice per workgroup, so that a batch of a few
  // hundred cylinders still fills the chip (80 workgroups walking 180k points each took 2.5 ms per call)
  const long long p_lo = range[c0 * 2], scene_hi = range[c0 * 2 + 1];
  const long long slice = ((scene_hi - p_lo + n_chunks - 1) / n_chunks + 255) / 256 * 256;
  const long long s_lo = p_lo + slice * blockIdx.y, p_hi = min(scene_hi, s_lo + slice);
  if (threadIdx.x < kCyl) {   // the cylinder's points in earlier slices precede this slice's in its list
    long long before = 0;
    if (starts && (int)threadIdx.x < nc) {
      before = starts[c0 + threadIdx.x];
      for (int ch = 0; ch < (int)blockIdx.y; ++ch) before += chunk_counts[(c0 + threadIdx.x) * n_chunks + ch];
    }
    chunk_base[threadIdx.x] = before;
  }
  __syncthreads();
  long long base[kCyl];
#pragma unroll
  for (int c = 0; c < kCyl; ++c) base[c] = chunk_base[c];
  int total[kCyl];
#pragma unroll
  for (int c = 0; c < kCyl; ++c) total[c] = 0;
  __syncthreads();
  for (long long p0 = s_lo; p0 < p_hi; p0 += 256) {
    const long long p = p0 + threadIdx.x;
    const bool live = p < p_hi;
    float x = 0.f, y = 0.f;
    bool recent = false;
    if (live) {
      const float* row = points + p * f;
      x = row[0];
      y = row[1];
      recent = time_col < 0 || row[time_col] < max_time;
    }
    unsigned long long bal[kCyl];
#pragma unroll
    for (int c = 0; c < kCyl; ++c) {
      const float dx = x - cx[c], dy = y - cy[c];
      const bool in = live && recent && sqrtf(dx * dx + dy * dy) <= cr[c];
      bal[c] = __ballot(in);
    }
    if (lane < kCyl) {
      unsigned long long mine = 0;
#pragma unroll
      for (int c = 0; c < kCyl; ++c) mine = (lane == c) ? bal[c] : mine;
      wave_cnt[wv][lane] = __popcll(mine);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kCyl; ++c) {
      int before = 0, all = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int n = wave_cnt[w][c];
        before += (w < wv) ? n : 0;
        all += n;
      }
      if (index && ((bal[c] >> lane) & 1ull))
        index[base[c] + total[c] + before + __popcll(bal[c] & ((1ull << lane) - 1ull))] = (int)(p - p_lo);
      total[c] += all;
    }
    __syncthreads();
  }
  if (counts && threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < kCyl; ++c)
      if (c < nc) {
        if (n_chunks == 1) {
          counts[c0 + c] = total[c];
        } else {
          chunk_counts[(c0 + c) * n_chunks + blockIdx.y] = total[c];
          if (total[c]) atomicAdd(&counts[c0 + c], total[c]);
        }
      }
  }
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_cylinder_select_f32(const float* points, int64_t n_points, int f, int time_col, float max_time,
                                       const int64_t* point_range, const float* centre_radius, int64_t n_cyl,
                                       const int64_t* starts, int32_t* counts, int32_t* index, int n_chunks,
                                       int32_t* chunk_counts, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(n_points >= 0 && f >= 2 && time_col < f && n_cyl >= 0, "cylinder_select: bad sizes");
  if (n_cyl == 0) return EFG_OK;
  EFG_CHECK_ARG(n_cyl % kCyl == 0, "cylinder_select: the cylinder list must be padded to a multiple of %d per scene", kCyl);
  EFG_CHECK_ARG(point_range && centre_radius && (counts || index) && (!index || starts) && (points || n_points == 0),
                "cylinder_select: null pointer");
  EFG_CHECK_ARG(n_chunks >= 1 && n_chunks <= 64 && (n_chunks == 1 || chunk_counts),
                "cylinder_select: 1 <= n_chunks <= 64, with a chunk_counts buffer when n_chunks > 1");
  static_assert(sizeof(long long) == sizeof(int64_t), "");
  hipLaunchKernelGGL(cylinder_select_kernel, dim3((unsigned)(n_cyl / kCyl), (unsigned)n_chunks), dim3(256), 0, stream, points, f,
                     time_col, max_time, reinterpret_cast<const long long*>(point_range), centre_radius, (long long)n_cyl,
                     reinterpret_cast<const long long*>(starts), counts, index, n_chunks, chunk_counts);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
