// Succinct rank index over a linearised cell grid: one uint2 {bits, prefix} per 32 cells, where
// prefix = number of set bits in all earlier words.  rank(cell) = prefix + popc(bits below cell)
// is the position of an active cell in ascending-cell ("canonical") order, so one structure
// gives (a) sorted compaction without a sort, (b) coordinate -> row lookup in ONE 8-byte load
// with no probing, (c) deterministic order.  Sized for big HBM: the 41x1504x1504 input grid of
// one scene is 2.9 M words = 23 MB.
//
// Used by dynamic scatter (voxel id = rank of the linearised coordinate, the order
// scatter_points_cuda.cu:237-251 gets from argsort+cumsum) and by the sparse-conv geometry.
#pragma once
#include "common.h"

namespace efg {

constexpr int kRankTileWords = 2048;  // words per block in the scan phases (256 threads x 8)

__device__ __forceinline__ int rank_lookup(const uint2* __restrict__ idx, unsigned long long cell) {
  const uint2 u = idx[cell >> 5];
  const unsigned bit = 1u << (cell & 31);
  if (!(u.x & bit)) return -1;
  return (int)(u.y + __popc(u.x & (bit - 1)));
}

__device__ __forceinline__ void rank_set(uint2* idx, unsigned long long cell) {
  atomicOr(&idx[cell >> 5].x, 1u << (cell & 31));
}

// as rank_set; true when this call turned the bit on (exactly one caller per cell sees true)
__device__ __forceinline__ bool rank_set_new(uint2* idx, unsigned long long cell) {
  const unsigned bit = 1u << (cell & 31);
  return !(atomicOr(&idx[cell >> 5].x, bit) & bit);
}

// phase a: popcount sum per tile of kRankTileWords words
static __global__ void __launch_bounds__(256) rank_tile_sums_kernel(const uint2* __restrict__ idx, long long words,
                                                              int* __restrict__ tile_sums) {
  __shared__ int smem[4];
  const long long base = (long long)blockIdx.x * kRankTileWords;
  int s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long long w = base + j * 256 + threadIdx.x;
    if (w < words) s += __popc(idx[w].x);
  }
  s = wave_reduce_sum(s);
  if (lane_id() == 0) smem[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = smem[0] + smem[1] + smem[2] + smem[3];
}

// phase b: exclusive scan of the tile sums by ONE block of 1024 threads; writes the grand total.
static __global__ void __launch_bounds__(1024) rank_scan_tiles_kernel(int* __restrict__ tile_sums, int ntiles,
                                                                int* __restrict__ total_out) {
  __shared__ int smem[17];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < ntiles) ? tile_sums[i] : 0;
    int tot;
    const int ex = block_exclusive_scan(v, smem, &tot);
    if (i < ntiles) tile_sums[i] = ex + carry;
    __syncthreads();
    if (threadIdx.x == 0) carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// phase c: per-word exclusive prefix.  Thread t of a tile owns 8 CONSECUTIVE words so that the
// in-thread running sum follows word order.
static __global__ void __launch_bounds__(256) rank_apply_kernel(uint2* __restrict__ idx, long long words,
                                                          const int* __restrict__ tile_sums) {
  __shared__ int smem[17];
  const long long base = (long long)blockIdx.x * kRankTileWords + threadIdx.x * 8;
  unsigned bits[8];
  int s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bits[j] = (base + j < words) ? idx[base + j].x : 0u;
    s += __popc(bits[j]);
  }
  int tot;
  int run = block_exclusive_scan(s, smem, &tot) + tile_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (base + j < words) idx[base + j].y = (unsigned)run;
    run += __popc(bits[j]);
  }
}

inline long long rank_words(unsigned long long cells) { return (long long)((cells + 31) / 32); }
inline int rank_tiles(long long words) { return (int)std::max<long long>(1, ceil_div(words, kRankTileWords)); }

// Enqueue phases a-c.  tile_sums: int[rank_tiles(words)] scratch.  total_dev may be NULL.
inline int rank_build_prefix(uint2* idx, long long words, int* tile_sums, int* total_dev, hipStream_t stream) {
  const int ntiles = rank_tiles(words);
  hipLaunchKernelGGL(rank_tile_sums_kernel, dim3(ntiles), dim3(256), 0, stream, idx, words, tile_sums);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(rank_scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, tile_sums, ntiles, total_dev);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(rank_apply_kernel, dim3(ntiles), dim3(256), 0, stream, idx, words, tile_sums);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

}  // namespace efg
