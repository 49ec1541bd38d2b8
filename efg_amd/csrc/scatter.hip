// Dynamic point -> voxel scatter-reduce for gfx950.
//
// Replaces efg::dynamic_point_to_voxel_forward / _backward
// (efg/operators/src/voxelize/voxelization.h:96-128; scatter_points_cuda.cu:209-352).
// The reference sorts int64 keys with ATen argsort, flags segment heads, cumsums and scatters
// the map back (5 ATen/own kernels + a host sync on the default stream).  Here the voxel id is
// the RANK of the linearised coordinate in a bitmap rank index (rank_index.h): identical
// ordering (ascending key), no sort, deterministic ids.
#include "rank_index.h"

namespace efg {
namespace {

constexpr int kMaxDim = 4;
struct Dims {
  int n;
  long long d[kMaxDim];
};

__device__ __forceinline__ long long coor_key(const int* __restrict__ c, const Dims& dm) {
  // scatter_points_cuda.cu:70-81: row-major id, -1 as soon as a coordinate is negative
  long long id = 0;
  for (int j = 0; j < dm.n; ++j) {
    const int t = c[j];
    if (t < 0) return -1;
    id = id * dm.d[j] + t;
  }
  return id;
}

__global__ void __launch_bounds__(256) scatter_mark_kernel(const int* __restrict__ coors, long long n, Dims dm,
                                                            uint2* __restrict__ idx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long key = coor_key(coors + i * dm.n, dm);
    if (key >= 0) rank_set(idx, (unsigned long long)key);
  }
}

__global__ void __launch_bounds__(256) scatter_map_kernel(const int* __restrict__ coors, long long n, Dims dm,
                                                           const uint2* __restrict__ idx, int* __restrict__ p2v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long key = coor_key(coors + i * dm.n, dm);
    p2v[i] = (key >= 0) ? rank_lookup(idx, (unsigned long long)key) : -1;
  }
}

__global__ void __launch_bounds__(256) fill_f32_kernel(float* __restrict__ p, long long n, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ void __launch_bounds__(256) fill_i32_kernel(int* __restrict__ p, long long n, int v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

// order-preserving float max through integer atomics (no CAS loop, cf. reduceMax
// scatter_points_cuda.cu:17-24)
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

// one thread per (point, channel): scatter_points_cuda.cu:101-133
__global__ void __launch_bounds__(256)
scatter_reduce_kernel(const float* __restrict__ feats, const int* __restrict__ coors, const int* __restrict__ p2v,
                      long long n, int c, int ndim, int reduce, float* __restrict__ vf, int* __restrict__ vc,
                      int* __restrict__ count) {
  const long long total = n * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / c;
    const int k = (int)(e - i * c);
    const int v = p2v[i];
    if (v < 0) continue;
    const float x = feats[e];
    if (reduce == 2) atomic_max_f32(vf + (long long)v * c + k, x);
    else unsafeAtomicAdd(vf + (long long)v * c + k, x);
    if (k == 0) {
      if (reduce == 1) atomicAdd(count + v, 1);
      for (int j = 0; j < ndim; ++j) vc[(long long)v * ndim + j] = coors[i * ndim + j];
    }
  }
}

// ---- deterministic sum / mean: exact integer accumulation ------------------------------------------------------
// Float atomics make the sum depend on the order the points arrive in (the reference's reduceAdd,
// scatter_points_cuda.cu:101-133, has the same property).  Here every addend is converted to a 64-bit fixed-point
// integer x * 2^s, with s chosen from the largest |x| of the call and the point count so that NO sum can overflow
// (n * max|x| * 2^s < 2^62); integer addition is associative, so the voxel sums are the same bits run to run,
// whatever the arrival order -- and they are the correctly rounded true sums, not a chain of fp32 roundings.
// x * 2^s is exact in double (24-bit mantissa times a power of two); bits of x below 2^-s are rounded to nearest.
// Non-finite addends (never produced by the pipeline) go through float atomics into the output itself: Inf / NaN
// results do not depend on the order either.
__global__ void __launch_bounds__(256) scatter_absmax_kernel(const float* __restrict__ feats, const int* __restrict__ p2v,
                                                              long long n, int c, unsigned* __restrict__ absmax_bits) {
  unsigned best = 0u;
  const long long total = n * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    if (p2v[e / c] < 0) continue;
    const float a = fabsf(feats[e]);
    if (a < INFINITY) best = max(best, __float_as_uint(a));   // (NaN fails the comparison; finite floats order like their bits)
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, d, 64));
  if (lane_id() == 0 && best) atomicMax(absmax_bits, best);
}

__device__ __forceinline__ int fixed_shift(unsigned absmax_bits, long long n) {
  int e = 0;
  frexpf(__uint_as_float(absmax_bits), &e);            // max|x| < 2^e (e = 0 for an all-zero call)
  int ln = 0;
  while ((1ll << ln) < n + 1) ++ln;                      // n + 1 <= 2^ln
  return min(61 - ln - e, 1000);
}

__global__ void __launch_bounds__(256)
scatter_sum_fixed_kernel(const float* __restrict__ feats, const int* __restrict__ coors, const int* __restrict__ p2v,
                         long long n, int c, int ndim, int reduce, const unsigned* __restrict__ absmax_bits,
                         unsigned long long* __restrict__ acc, float* __restrict__ vf, int* __restrict__ vc,
                         int* __restrict__ count) {
  const int s = fixed_shift(*absmax_bits, n);
  const long long total = n * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / c;
    const int k = (int)(e - i * c);
    const int v = p2v[i];
    if (v < 0) continue;
    const float x = feats[e];
    if (fabsf(x) < INFINITY)
      atomicAdd(acc + (long long)v * c + k, (unsigned long long)__double2ll_rn(ldexp((double)x, s)));   // two's complement
    else
      unsafeAtomicAdd(vf + (long long)v * c + k, x);
    if (k == 0) {
      if (reduce == 1) atomicAdd(count + v, 1);
      for (int j = 0; j < ndim; ++j) vc[(long long)v * ndim + j] = coors[i * ndim + j];
    }
  }
}

__global__ void __launch_bounds__(256)
scatter_fixed_finish_kernel(const unsigned long long* __restrict__ acc, const unsigned* __restrict__ absmax_bits,
                            long long n, const int* __restrict__ count, long long m, int c, int reduce,
                            float* __restrict__ vf) {
  const int s = fixed_shift(*absmax_bits, n);
  const long long total = m * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    // the exact sum, rounded ONCE to fp32 (+ the non-finite part, 0 in every sane call)
    float v = __fadd_rn(vf[e], (float)ldexp((double)(long long)acc[e], -s));
    if (reduce == 1) v = __fdiv_rn(v, (float)count[e / c]);
    vf[e] = v;
  }
}

__global__ void __launch_bounds__(256) scatter_mean_kernel(float* __restrict__ vf, const int* __restrict__ count,
                                                            long long m, int c) {
  const long long total = m * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x)
    vf[e] = __fdiv_rn(vf[e], (float)count[e / c]);
}

// scatter_points_cuda.cu:136-162
__global__ void __launch_bounds__(256)
scatter_bwd_add_kernel(float* __restrict__ gf, const float* __restrict__ gv, const int* __restrict__ p2v,
                       const int* __restrict__ count, long long n, int c, int reduce) {
  const long long total = n * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / c;
    const int k = (int)(e - i * c);
    const int v = p2v[i];
    float g = 0.0f;
    if (v >= 0) {
      g = gv[(long long)v * c + k];
      if (reduce == 1) g = __fdiv_rn(g, (float)count[v]);
    }
    gf[e] = g;
  }
}

// scatter_points_cuda.cu:165-186: lowest point index attaining the max, per (voxel, channel)
__global__ void __launch_bounds__(256)
scatter_bwd_argmax_kernel(const float* __restrict__ feats, const float* __restrict__ vf, const int* __restrict__ p2v,
                          long long n, int c, int* __restrict__ from) {
  const long long total = n * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / c;
    const int k = (int)(e - i * c);
    const int v = p2v[i];
    if (v < 0) continue;
    if (feats[e] == vf[(long long)v * c + k]) atomicMin(from + (long long)v * c + k, (int)i);
  }
}

// scatter_points_cuda.cu:189-205
__global__ void __launch_bounds__(256)
scatter_bwd_max_kernel(float* __restrict__ gf, const float* __restrict__ gv, const int* __restrict__ from,
                       long long m, int c, long long n) {
  const long long total = m * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int src = from[e];
    if ((long long)src < n) gf[(long long)src * c + (e % c)] = gv[e];
  }
}

int make_dims(int ndim, const int32_t* dims_host, Dims* dm, unsigned long long* cells) {
  EFG_CHECK_ARG(ndim >= 1 && ndim <= kMaxDim, "scatter: coors must have 1..%d columns, got %d", kMaxDim, ndim);
  dm->n = ndim;
  unsigned long long c = 1;
  for (int j = 0; j < ndim; ++j) {
    const long long d = std::max(dims_host[j], 0);
    dm->d[j] = d;
    c *= (unsigned long long)d;
    EFG_CHECK_ARG(c < 0xffffffffull, "scatter: coordinate space too large (>= 2^32 cells)");
  }
  *cells = c;
  return EFG_OK;
}

inline int grid_for(long long work) { return (int)std::min<long long>(std::max<long long>(ceil_div(work, 256), 1), 4096); }

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" size_t efg_scatter_workspace_bytes(int64_t n, int ndim, const int32_t* dims_host) {
  Dims dm;
  unsigned long long cells;
  if (make_dims(ndim, dims_host, &dm, &cells) != EFG_OK) return 0;
  const long long words = std::max<long long>(rank_words(cells), 1);
  return align_up((size_t)words * 8, 256) + align_up((size_t)rank_tiles(words) * 4, 256) + 256;
}

extern "C" int efg_scatter_index(const int32_t* coors, int64_t n, int ndim, const int32_t* dims_host,
                                 int32_t* point2voxel, int32_t* m_dev, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Dims dm;
  unsigned long long cells;
  if (int rc = make_dims(ndim, dims_host, &dm, &cells)) return rc;
  EFG_CHECK_ARG(n >= 0 && n < (1ll << 31), "scatter: bad point count");
  const long long words = std::max<long long>(rank_words(cells), 1);
  Workspace w(ws, ws_bytes);
  uint2* idx = w.take<uint2>(words);
  int* tile_sums = w.take<int>(rank_tiles(words));
  if (!w.ok) {
    set_error("scatter workspace too small: need %zu bytes, got %zu", efg_scatter_workspace_bytes(n, ndim, dims_host),
              ws_bytes);
    return EFG_E_WORKSPACE;
  }
  EFG_HIP_TRY(hipMemsetAsync(idx, 0, (size_t)words * 8, stream));
  if (n > 0) {
    hipLaunchKernelGGL(scatter_mark_kernel, dim3(grid_for(n)), dim3(256), 0, stream, coors, (long long)n, dm, idx);
    EFG_LAUNCH_CHECK();
  }
  if (int rc = rank_build_prefix(idx, words, tile_sums, m_dev, stream)) return rc;
  if (n > 0) {
    hipLaunchKernelGGL(scatter_map_kernel, dim3(grid_for(n)), dim3(256), 0, stream, coors, (long long)n, dm, idx,
                       point2voxel);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

extern "C" size_t efg_scatter_reduce_workspace_bytes(int64_t m, int c) {
  if (m < 0 || c < 1) return 0;
  return align_up((size_t)std::max<int64_t>(m, 1) * c * 8, 256) + 256;   // int64 accumulators + the |x| maximum
}

extern "C" int efg_scatter_reduce_f32(const float* feats, const int32_t* coors, const int32_t* p2v, int64_t n, int c,
                                      int ndim, int reduce, int64_t m, float* vf, int32_t* vc, int32_t* count,
                                      void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(reduce >= 0 && reduce <= 2, "scatter: reduce must be 0 (sum), 1 (mean) or 2 (max)");
  EFG_CHECK_ARG(c >= 1 && ndim >= 1 && ndim <= kMaxDim && n >= 0 && m >= 0, "scatter: bad sizes");
  if (m == 0) return EFG_OK;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(m * c)), dim3(256), 0, stream, vf, (long long)m * c,
                     reduce == 2 ? -INFINITY : 0.0f);
  EFG_LAUNCH_CHECK();
  EFG_HIP_TRY(hipMemsetAsync(count, 0, (size_t)m * 4, stream));
  if (reduce == 2) {   // max: order-independent as it is
    if (n > 0) {
      hipLaunchKernelGGL(scatter_reduce_kernel, dim3(grid_for((long long)n * c)), dim3(256), 0, stream, feats, coors,
                         p2v, (long long)n, c, ndim, reduce, vf, vc, count);
      EFG_LAUNCH_CHECK();
    }
    return EFG_OK;
  }
  // sum / mean: exact integer accumulation (deterministic; see scatter_sum_fixed_kernel)
  Workspace w(ws, ws_bytes);
  unsigned long long* acc = w.take<unsigned long long>((size_t)m * c);
  unsigned* absmax = w.take<unsigned>(1);
  if (!w.ok) {
    set_error("scatter reduce workspace too small: need %zu bytes, got %zu", efg_scatter_reduce_workspace_bytes(m, c), ws_bytes);
    return EFG_E_WORKSPACE;
  }
  EFG_HIP_TRY(hipMemsetAsync(acc, 0, reinterpret_cast<char*>(absmax + 1) - reinterpret_cast<char*>(acc), stream));
  if (n > 0) {
    hipLaunchKernelGGL(scatter_absmax_kernel, dim3(grid_for((long long)n * c)), dim3(256), 0, stream, feats, p2v, (long long)n,
                       c, absmax);
    hipLaunchKernelGGL(scatter_sum_fixed_kernel, dim3(grid_for((long long)n * c)), dim3(256), 0, stream, feats, coors, p2v,
                       (long long)n, c, ndim, reduce, absmax, acc, vf, vc, count);
    EFG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(scatter_fixed_finish_kernel, dim3(grid_for(m * c)), dim3(256), 0, stream, acc, absmax, (long long)n, count,
                     (long long)m, c, reduce, vf);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_scatter_backward_f32(float* gf, const float* gv, const float* feats, const float* vf,
                                        const int32_t* p2v, const int32_t* count, int64_t n, int64_t m, int c,
                                        int reduce, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(reduce >= 0 && reduce <= 2, "scatter: reduce must be 0 (sum), 1 (mean) or 2 (max)");
  if (n == 0) return EFG_OK;
  if (reduce != 2) {
    hipLaunchKernelGGL(scatter_bwd_add_kernel, dim3(grid_for((long long)n * c)), dim3(256), 0, stream, gf, gv, p2v,
                       count, (long long)n, c, reduce);
    EFG_LAUNCH_CHECK();
    return EFG_OK;
  }
  EFG_HIP_TRY(hipMemsetAsync(gf, 0, (size_t)n * c * 4, stream));
  if (m == 0) return EFG_OK;
  if (ws_bytes < (size_t)m * c * 4) {
    set_error("scatter backward (max) workspace too small: need %zu bytes", (size_t)m * c * 4);
    return EFG_E_WORKSPACE;
  }
  int* from = static_cast<int*>(ws);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(m * c)), dim3(256), 0, stream, from, (long long)m * c, 0x7fffffff);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(scatter_bwd_argmax_kernel, dim3(grid_for((long long)n * c)), dim3(256), 0, stream, feats, vf, p2v,
                     (long long)n, c, from);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(scatter_bwd_max_kernel, dim3(grid_for(m * c)), dim3(256), 0, stream, gf, gv, from, (long long)m,
                     c, (long long)n);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
