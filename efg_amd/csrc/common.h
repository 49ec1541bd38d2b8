// Shared host/device helpers for libefg_hip.so (gfx950 only: wave64, no portability layer).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "efg_hip.h"

namespace efg {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define EFG_CHECK_ARG(cond, ...)   \
  do {                             \
    if (!(cond)) {                 \
      efg::set_error(__VA_ARGS__); \
      return EFG_E_INVALID;        \
    }                              \
  } while (0)

#define EFG_HIP_TRY(expr)                                                                    \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      efg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return EFG_E_HIP;                                                                      \
    }                                                                                        \
  } while (0)

#define EFG_LAUNCH_CHECK() EFG_HIP_TRY(hipGetLastError())

// Opt a kernel into more than 64 KB of dynamic LDS, once per DEVICE (function attributes are per device; the flag
// word is per call site).  One process normally drives one GPU, but the C ABI accepts pointers on any device.
#define EFG_ALLOW_DYNAMIC_LDS(kernel, bytes)                                                              \
  do {                                                                                                    \
    static std::atomic<unsigned long long> efg_lds_done_{0};                                              \
    int efg_dev_ = 0;                                                                                     \
    EFG_HIP_TRY(hipGetDevice(&efg_dev_));                                                                 \
    const unsigned long long efg_bit_ = 1ull << (efg_dev_ & 63);                                          \
    if (!(efg_lds_done_.load(std::memory_order_relaxed) & efg_bit_)) {                                    \
      EFG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));         \
      efg_lds_done_.fetch_or(efg_bit_, std::memory_order_relaxed);                                        \
    }                                                                                                     \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over the caller-provided workspace.
struct Workspace {
  char* base;
  size_t cap;
  size_t off = 0;
  bool ok = true;
  Workspace(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T), 256);
    if (off + bytes > cap) {
      ok = false;
      return nullptr;
    }
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
};

// ---- device helpers ---------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane_id() >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ int wave_reduce_sum(int v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Block-wide exclusive scan of one int per thread (blockDim.x multiple of 64, <= 1024).
// Returns the exclusive prefix; *total receives the block sum.  smem: int[17].
__device__ __forceinline__ int block_exclusive_scan(int v, int* smem, int* total) {
  const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int inc = wave_inclusive_scan(v);
  if (lane_id() == 63) smem[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int s = (lane_id() < nw) ? smem[lane_id()] : 0;
    int si = wave_inclusive_scan(s);
    if (lane_id() < nw) smem[lane_id()] = si - s;
    if (lane_id() == nw - 1) smem[16] = si;
  }
  __syncthreads();
  const int r = inc - v + smem[wid];
  *total = smem[16];
  __syncthreads();
  return r;
}

// `n` consecutive words of a per-device ring of device words that are ZERO at rest (csrc/colsum.hip): ticket counters and
// small scratch of single-launch reductions -- whoever draws them stores 0 back before its kernel ends.  nullptr while the
// stream is being captured (a slot baked into a graph could meet an eager call's) or when the ring cannot be created: the
// caller then takes its form without tickets.
unsigned* ticket_slots(int n, hipStream_t st);

}  // namespace efg
