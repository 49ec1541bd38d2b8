// Unsorted top-k of every row of a [rows, n] fp32 matrix, for gfx950: exact radix select + ordered compaction.
//
// Replaces the torch.topk(..., sorted=False) of the proposal selection ($CQ/transformer.py:65: the num_queries best of
// the 35 344 encoder tokens per scene).  ATen's multi-block radix select is 21 launches (~100 us) and returns ANY members
// of a tie at the cut; on the plateau of equal scores a freshly initialised model produces (every empty BEV cell has
// the same logit) two runs of one step pick different proposals.  Here: ONE launch, one workgroup per row; four 8-bit
// digit passes over the order-preserving integer image of the floats find the k-th largest value exactly, then an
// ordered compaction takes every larger element and, of the elements EQUAL to it, the ones with the lowest indices.
// Output in ascending index order (the reference's order is unspecified: sorted=False).
#include "common.h"

namespace efg {
namespace {

__device__ __forceinline__ unsigned order_key(float x) {
  const unsigned b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // larger float <=> larger key (NaN sorts above +Inf, like torch)
}

__device__ __forceinline__ float key_value(unsigned k) {   // inverse of order_key
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// STAGED: the row's keys are read ONCE into LDS (up to 36 864 elements = 144 KB; the proposal selection's 35 344 fit) and the
// four digit passes and the two compaction passes read them there.  Unstaged, every pass walks the row in global memory
// with one load in flight per thread: 6 x 35 dependent round trips were the kernel's 86 us.
template <bool STAGED>
__global__ void __launch_bounds__(1024) topk_kernel(const float* __restrict__ x, int n, int k, float* __restrict__ values,
                                                     long long* __restrict__ indices) {
  extern __shared__ unsigned staged_keys[];
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need;
  __shared__ int smem[17];
  const float* row = x + (long long)blockIdx.x * n;
  const int tid = threadIdx.x;
  if (STAGED) {
    for (int i0 = tid; i0 < n; i0 += 8 * 1024) {   // eight loads in flight per thread
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = row[min(i0 + u * 1024, n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * 1024 < n) staged_keys[i0 + u * 1024] = order_key(v[u]);
    }
    __syncthreads();
  }
  auto key_at = [&](int i) { return STAGED ? staged_keys[i] : order_key(row[i]); };
  unsigned prefix = 0u, need = (unsigned)k;   // `need` = how many of the elements matching `prefix` so far are still wanted
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    // Scores cluster (a fresh model puts every token near the prior: one or two digits hold the whole row), and 64 lanes
    // adding to ONE LDS counter are 64 serial operations.  A wave first counts, up to three times, the lanes that share
    // the digit of its first pending lane (ballot + popcount, one atomic for all of them); what is left (spread-out data)
    // adds itself.
    for (int i0 = 0; i0 < n; i0 += 1024) {
      const int i = i0 + tid;
      unsigned key = 0u;
      bool pending = i < n;
      if (pending) {
        key = key_at(i);
        pending = pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8));
      }
      const unsigned digit = (key >> shift) & 255u;
#pragma unroll
      for (int round = 0; round < 3; ++round) {
        const unsigned long long todo = __ballot(pending);
        if (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const unsigned dl = (unsigned)__shfl((int)digit, leader, 64);
          const unsigned long long same = __ballot(pending && digit == dl);
          if ((tid & 63) == leader) atomicAdd(&hist[dl], (unsigned)__popcll(same));
          if (digit == dl) pending = false;
        }
      }
      if (pending) atomicAdd(&hist[digit], 1u);
    }
    __syncthreads();
    if (tid < 64) {   // one wave: the digit d with  (count of larger digits) < need <= (count of digits >= d)
      unsigned c[4], above = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = hist[255 - (tid * 4 + q)];   // lane 0 holds the four LARGEST digits
      const unsigned mine = c[0] + c[1] + c[2] + c[3];
      unsigned inc = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(inc, d, 64);
        if (tid >= d) inc += t;
      }
      above = inc - mine;   // elements in digits larger than this lane's four
      if (above < need && need <= inc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (above < need && need <= above + c[q]) {
            s_prefix = prefix | ((unsigned)(255 - (tid * 4 + q)) << shift);
            s_need = need - above;
          }
          above += c[q];
        }
      }
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    __syncthreads();
  }
  // prefix = key of the k-th largest element; `need` of the elements equal to it are taken, lowest indices first
  const int per = (n + 1023) / 1024, i0 = min(tid * per, n), i1 = min(i0 + per, n);
  int gt = 0, eq = 0;
  for (int i = i0; i < i1; ++i) {
    const unsigned key = key_at(i);
    gt += key > prefix ? 1 : 0;
    eq += key == prefix ? 1 : 0;
  }
  int tot;
  int gt_before = block_exclusive_scan(gt, smem, &tot);
  int eq_before = block_exclusive_scan(eq, smem, &tot);
  float* vo = values + (long long)blockIdx.x * k;
  long long* io = indices + (long long)blockIdx.x * k;
  for (int i = i0; i < i1; ++i) {
    const unsigned key = key_at(i);
    const float v = STAGED ? key_value(key) : row[i];
    const bool take = key > prefix || (key == prefix && eq_before < (int)need);
    if (take) {
      const int pos = gt_before + min(eq_before, (int)need);
      vo[pos] = v;
      io[pos] = i;
    }
    gt_before += key > prefix ? 1 : 0;
    eq_before += key == prefix ? 1 : 0;
  }
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_topk_unsorted_f32(const float* x, int64_t rows, int n, int k, float* values, int64_t* indices,
                                     void* stream) {
  EFG_CHECK_ARG(rows >= 0 && n >= 1 && k >= 1 && k <= n, "topk: need 1 <= k <= n (k=%d, n=%d)", k, n);
  EFG_CHECK_ARG(rows < (1ll << 31), "topk: too many rows");
  if (rows == 0) return EFG_OK;
  constexpr int kStagedMax = 36864;   // 144 KB of keys beside the histogram
  if (n <= kStagedMax) {
    EFG_ALLOW_DYNAMIC_LDS(topk_kernel<true>, kStagedMax * 4);   // (once per device; common.h)
    hipLaunchKernelGGL(topk_kernel<true>, dim3((unsigned)rows), dim3(1024), (size_t)n * 4, (hipStream_t)stream, x, n, k, values,
                       reinterpret_cast<long long*>(indices));
  } else {
    hipLaunchKernelGGL(topk_kernel<false>, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, x, n, k, values,
                       reinterpret_cast<long long*>(indices));
  }
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
