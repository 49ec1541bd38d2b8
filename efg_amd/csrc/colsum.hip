// Column sums of a row-major fp32 matrix for gfx950: the bias gradient of the path's Linear layers.
//
// Every nn.Linear of the transformer ($CQ/transformer.py:215-243,273-317 linear1 / linear2, the MLP heads
// $CQ/modules/blocks.py:5-17, $CQ/modules/box_attention.py:31-40 value_proj / output_proj / linear_box / linear_attn)
// gets its bias gradient from autograd as grad_output.sum(0).  A training step holds ~100 of those reductions:
// 24 over the 70 688-token encoder sequence (72-290 MB each, run by ATen at 1.8 TB/s) and ~70 over a few thousand
// decoder rows (15 us each for 2.5 MB: launch-shaped, not bandwidth-shaped) -- 1.7 ms per step together.
//
//   K1 partial  one workgroup per block of rows; a lane owns one float4 (or scalar) column slot and, for narrow
//               matrices, a wave covers several rows per step (64 / pow2ceil(columns/4)); row loop unrolled so
//               that each wave keeps several KB in flight; rows -> waves -> LDS, fixed order;
//   K2 final    partial[blocks][C] -> out[C], lanes over columns (coalesced), waves over blocks, fixed order.
// Up to kFusedBlocks row blocks (the decoder's few thousand rows) K2 runs INSIDE K1: every block publishes its partial row
// and takes a ticket; the block that draws the last one sums the partial rows in block order (the ticket decides WHO sums,
// not the order) -- one launch instead of two for ~70 reductions of a step.  The tickets live in a ring of self-resetting
// device counters (a call takes the next slots; the summing block stores 0 back).
// Deterministic (no float atomics).  HBM-bound: one pass over the matrix.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace efg {
namespace {

constexpr int kMaxBlocks = 512;

template <bool kVec>
struct Acc;
template <>
struct Acc<true> {
  float4 v;
  __device__ void zero() { v = float4{0.f, 0.f, 0.f, 0.f}; }
  __device__ void add_from(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  __device__ void add(const Acc& o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
  __device__ Acc xor_lane(int m) const {
    Acc r;
    r.v = float4{__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64), __shfl_xor(v.w, m, 64)};
    return r;
  }
  __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
};
template <>
struct Acc<false> {
  float v;
  __device__ void zero() { v = 0.f; }
  __device__ void add_from(const float* p) { v += *p; }
  __device__ void add(const Acc& o) { v += o.v; }
  __device__ Acc xor_lane(int m) const {
    Acc r;
    r.v = __shfl_xor(v, m, 64);
    return r;
  }
  __device__ void store(float* p) const { *p = v; }
};

constexpr int kFusedBlocks = 128;   // row blocks up to which the final sum runs inside the partial kernel

// Called by every thread of a block after its partial row is stored (by lanes of wave 0): true in the one block that has
// to sum.  Release/acquire at agent scope: the partial rows were written through other XCDs' L2s.
// A counter that was NOT at rest when its launch began (a launch that died half way through its tickets -- normally the
// end of the HIP context, but nothing guarantees it) would make the wrong block sum, over rows that are not all written;
// the symptom this can see -- a ticket beyond the launch's block count -- is counted here and reported by
// efg_ticket_ring_errors (the trainer's periodic anomaly check reads it), and the slot is put back to rest.
__device__ unsigned g_ticket_errors;

__device__ __forceinline__ bool draw_last_ticket(unsigned* counter, unsigned nblocks) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev >= nblocks) {
      atomicAdd(&g_ticket_errors, 1u);
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = prev == nblocks - 1;
    if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for its next call
  }
  __syncthreads();
  return s_last != 0;
}

// out[c] = sum_b partial[b][c] for the columns [c0, c0 + ncols) of this block's column group, blocks in index order
// (four interleaved chains, combined in a fixed order).
__device__ __forceinline__ void sum_partials_in_block(const float* __restrict__ partial, int nblocks, int cols, int c0,
                                                      int ncols, float* __restrict__ out) {
  for (int t = threadIdx.x; t < ncols; t += blockDim.x) {
    const int c = c0 + t;
    if (c >= cols) break;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 3 < nblocks; b += 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] += partial[(long long)(b + i) * cols + c];
    }
    for (; b < nblocks; ++b) s[0] += partial[(long long)b * cols + c];
    out[c] = (s[0] + s[1]) + (s[2] + s[3]);
  }
}

// slots = number of column slots (C / 4 float4 or C scalars); slot_log2 = log2(pow2ceil(min(slots, 64))).
template <bool kVec>
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ x, long long rows, int slots, long long row_stride,
                      int rows_per_block, int slot_log2, int cols, float* __restrict__ partial,
                      unsigned* __restrict__ tickets, float* __restrict__ out) {
  constexpr int kW = kVec ? 4 : 1;
  __shared__ Acc<kVec> sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int width = 1 << slot_log2;  // lanes that share a row
  const int rps = 64 >> slot_log2;   // rows a wave covers per step
  const int slot = blockIdx.y * 64 + (lane & (width - 1));
  const int rsub = lane >> slot_log2;
  const bool ok = slot < slots;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(rows, r0 + rows_per_block);
  Acc<kVec> a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i].zero();
  if (ok) {
    const float* base = x + (long long)slot * kW;
    const long long step = 4ll * rps;
    long long r = r0 + (long long)wave * rps + rsub;
    for (; r + 7 * step < r1; r += 8 * step) {  // eight independent loads in flight per lane
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i].add_from(base + (r + i * step) * row_stride);
    }
    for (; r < r1; r += step) a[0].add_from(base + r * row_stride);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i].add(a[i + 4]);
  a[0].add(a[2]);
  a[1].add(a[3]);
  a[0].add(a[1]);
  Acc<kVec>& a0 = a[0];
  for (int m = width; m < 64; m <<= 1) a0.add(a0.xor_lane(m));  // the wave's row groups
  if (rsub == 0) sm[wave][lane] = a0;
  __syncthreads();
  if (wave == 0 && rsub == 0 && ok) {
    Acc<kVec> s = sm[0][lane];
    s.add(sm[1][lane]);
    s.add(sm[2][lane]);
    s.add(sm[3][lane]);
    s.store(partial + (long long)blockIdx.x * cols + (long long)slot * kW);
  }
  if (tickets && draw_last_ticket(tickets + blockIdx.y, gridDim.x))
    sum_partials_in_block(partial, gridDim.x, cols, blockIdx.y * 64 * kW, 64 * kW, out);
}

// The same pass with the ReLU backward folded in: g_out = g where y > 0 else 0 (what autograd's threshold_backward
// writes), column sums of g_out in the order of colsum_partial_kernel<true> -- the masked gradient of a Linear + ReLU is
// written once and never re-read for its bias gradient (one 290 MB read less per FFN layer of the encoder).
__global__ void __launch_bounds__(256)
relu_bwd_colsum_partial_kernel(const float* __restrict__ g, const float* __restrict__ y, long long rows, int slots,
                               long long row_stride, int rows_per_block, int slot_log2, int cols, float* __restrict__ g_out,
                               float* __restrict__ partial, unsigned* __restrict__ tickets, float* __restrict__ out) {
  __shared__ Acc<true> sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int width = 1 << slot_log2;
  const int rps = 64 >> slot_log2;
  const int slot = blockIdx.y * 64 + (lane & (width - 1));
  const int rsub = lane >> slot_log2;
  const bool ok = slot < slots;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(rows, r0 + rows_per_block);
  Acc<true> a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i].zero();
  auto take = [&](Acc<true>& acc, long long r) {
    const long long at = r * row_stride + (long long)slot * 4;
    float4 t = *reinterpret_cast<const float4*>(g + at);
    const float4 m = *reinterpret_cast<const float4*>(y + at);
    t.x = m.x > 0.f ? t.x : 0.f;
    t.y = m.y > 0.f ? t.y : 0.f;
    t.z = m.z > 0.f ? t.z : 0.f;
    t.w = m.w > 0.f ? t.w : 0.f;
    *reinterpret_cast<float4*>(g_out + at) = t;
    acc.v.x += t.x; acc.v.y += t.y; acc.v.z += t.z; acc.v.w += t.w;
  };
  if (ok) {
    const long long step = 4ll * rps;
    long long r = r0 + (long long)wave * rps + rsub;
    for (; r + 7 * step < r1; r += 8 * step) {
#pragma unroll
      for (int i = 0; i < 8; ++i) take(a[i], r + i * step);
    }
    for (; r < r1; r += step) take(a[0], r);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i].add(a[i + 4]);
  a[0].add(a[2]);
  a[1].add(a[3]);
  a[0].add(a[1]);
  Acc<true>& a0 = a[0];
  for (int m = width; m < 64; m <<= 1) a0.add(a0.xor_lane(m));
  if (rsub == 0) sm[wave][lane] = a0;
  __syncthreads();
  if (wave == 0 && rsub == 0 && ok) {
    Acc<true> s = sm[0][lane];
    s.add(sm[1][lane]);
    s.add(sm[2][lane]);
    s.add(sm[3][lane]);
    s.store(partial + (long long)blockIdx.x * cols + (long long)slot * 4);
  }
  if (tickets && draw_last_ticket(tickets + blockIdx.y, gridDim.x))
    sum_partials_in_block(partial, gridDim.x, cols, blockIdx.y * 256, 256, out);
}

// The same for matrices whose rows are whole multiples of 4 KB (the 1024-wide FFN hidden gradient): thread t of the workgroup
// owns the float4 slot  blockIdx.y * 256 + t  of every row, so ONE load instruction of the workgroup reads 4 contiguous KB
// of a row (the kernel above reads 1 KB per wave from four different rows: 2.8 TB/s on the 3 x 290 MB of an encoder FFN
// layer) and no cross-wave reduction is needed: a thread's eight interleaved row accumulators are its column sums.
template <bool RELU>   // RELU: g_out = g * (y > 0) is written and summed; else the column sums of g alone (y, g_out unused)
__global__ void __launch_bounds__(256)
colsum_rowwide_kernel(const float* __restrict__ g, const float* __restrict__ y, long long rows, long long row_stride,
                      int rows_per_block, int cols, float* __restrict__ g_out, float* __restrict__ partial,
                      unsigned* __restrict__ tickets, float* __restrict__ out) {
  const int slot = blockIdx.y * 256 + threadIdx.x;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(rows, r0 + rows_per_block);
  Acc<true> a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i].zero();
  auto take = [&](Acc<true>& acc, long long r) {
    const long long at = r * row_stride + (long long)slot * 4;
    float4 t = *reinterpret_cast<const float4*>(g + at);
    if constexpr (RELU) {
      const float4 m = *reinterpret_cast<const float4*>(y + at);
      t.x = m.x > 0.f ? t.x : 0.f;
      t.y = m.y > 0.f ? t.y : 0.f;
      t.z = m.z > 0.f ? t.z : 0.f;
      t.w = m.w > 0.f ? t.w : 0.f;
      *reinterpret_cast<float4*>(g_out + at) = t;
    }
    acc.v.x += t.x; acc.v.y += t.y; acc.v.z += t.z; acc.v.w += t.w;
  };
  long long r = r0;
  for (; r + 7 < r1; r += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) take(a[i], r + i);
  }
  for (; r < r1; ++r) take(a[0], r);
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i].add(a[i + 4]);
  a[0].add(a[2]);
  a[1].add(a[3]);
  a[0].add(a[1]);
  a[0].store(partial + (long long)blockIdx.x * cols + (long long)slot * 4);
  if (tickets && draw_last_ticket(tickets + blockIdx.y, gridDim.x))
    sum_partials_in_block(partial, gridDim.x, cols, blockIdx.y * 1024, 1024, out);
}

// kWaves = 16 for long partial lists (the 70 688-row matrices), 4 for the decoder-sized ones.
template <int kWaves>
__global__ void __launch_bounds__(64 * kWaves)
colsum_final_kernel(const float* __restrict__ partial, int nblocks, int cols, float* __restrict__ out) {
  __shared__ float sm[kWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    int b = wave;
    for (; b + 3 * kWaves < nblocks; b += 4 * kWaves) {  // 4 independent loads in flight
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] += partial[(long long)(b + kWaves * i) * cols + c];
    }
    for (; b < nblocks; b += kWaves) s[0] += partial[(long long)b * cols + c];
  }
  sm[wave][lane] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (wave == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) t += sm[w][lane];
    out[c] = t;
  }
}

struct ColsumPlan {
  bool vec;
  int slots, slot_log2, rows_per_block, nblocks, ygroups;
};

ColsumPlan colsum_plan(int64_t rows, int cols, int64_t row_stride, const void* x) {
  ColsumPlan p;
  p.vec = (cols % 4 == 0) && (row_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.slots = p.vec ? cols / 4 : cols;
  int lg = 0;
  while ((1 << lg) < std::min(p.slots, 64)) ++lg;
  p.slot_log2 = lg;
  const int rps = 64 >> lg;
  // >= 8 row steps per wave (one fully unrolled batch of loads) per block, and no more than kMaxBlocks partial rows
  const int64_t min_rows = 32ll * rps;
  p.rows_per_block = (int)std::max<int64_t>(min_rows, ceil_div(std::max<int64_t>(rows, 1), kMaxBlocks));
  p.nblocks = (int)std::max<int64_t>(1, ceil_div(std::max<int64_t>(rows, 1), p.rows_per_block));
  p.ygroups = (int)ceil_div(p.slots, 64);
  return p;
}

// Rows of whole 4 KB pieces (contiguous, 16-byte aligned): the row-wide kernels -- a workgroup reads contiguous rows -- with as
// many row blocks as keep every CU busy twice over, each at least 64 rows long.  (A/B against the strip kernels on the
// [70 688, 1024] hidden gradient of an encoder FFN layer: 220 -> 174 us, 3.94 -> 4.99 TB/s; profiles/r06_colsum_rowwide_ab.txt.)
bool rowwide_plan(int64_t rows, int cols, int64_t row_stride, const void* x, ColsumPlan* p) {
  if (cols % 1024 != 0 || row_stride != cols || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return false;
  p->rows_per_block = (int)std::max<int64_t>(64, ceil_div(rows, kMaxBlocks));
  p->nblocks = (int)std::max<int64_t>(1, ceil_div(rows, p->rows_per_block));
  p->ygroups = cols / 1024;
  return true;
}

}  // namespace

// Ticket counters of the fused final: a ring of zero-initialised device words per device; a call takes `n` consecutive ones
// (one per column group).  A slot comes round again after kRing / n calls -- far beyond what a stream keeps in flight --
// and each counter is back at 0 when its kernel ends.  nullptr (=> two launches) while the stream is being captured.
namespace {
constexpr unsigned kRing = 8192;
std::mutex g_ring_mu;
unsigned* g_ring[64] = {};
unsigned g_ring_next[64] = {};
}  // namespace

unsigned* ticket_slots(int n, hipStream_t st) {   // (declared in common.h: det_loss.hip draws from the same ring)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || n < 1 || (unsigned)n > kRing / 4) return nullptr;
  // a launch that is being CAPTURED keeps the two-launch form: its ring slot would be baked into the graph and every
  // replay would use it again, possibly beside an eager call (or another graph) that drew the same slot
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st && (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) return nullptr;   // (the null stream cannot capture)
  std::lock_guard<std::mutex> lock(g_ring_mu);
  if (!g_ring[dev]) {
    unsigned* p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned) * kRing) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, sizeof(unsigned) * kRing) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    g_ring[dev] = p;
  }
  if (g_ring_next[dev] + (unsigned)n > kRing) g_ring_next[dev] = 0;
  unsigned* out = g_ring[dev] + g_ring_next[dev];
  g_ring_next[dev] += (unsigned)n;
  return out;
}

}  // namespace efg

using namespace efg;

extern "C" size_t efg_colsum_workspace_bytes(int64_t rows, int cols) {
  if (rows < 0 || cols < 1) return 0;
  // a block covers >= 16 rows (colsum_plan), whatever the alignment of the view turns out to be
  const int64_t nblocks = std::min<int64_t>(kMaxBlocks, std::max<int64_t>(1, ceil_div(rows, 16)));
  return align_up(sizeof(float) * (size_t)cols * (size_t)nblocks, 256);
}

extern "C" int efg_colsum_f32(const float* x, int64_t rows, int cols, int64_t row_stride, float* out, void* ws,
                              size_t ws_bytes, void* stream) {
  EFG_CHECK_ARG(rows >= 0 && cols >= 1, "colsum: bad shape %lld x %d", (long long)rows, cols);
  EFG_CHECK_ARG(row_stride >= cols, "colsum: row_stride %lld < cols %d", (long long)row_stride, cols);
  EFG_CHECK_ARG(out, "colsum: null output");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    EFG_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * cols, st));
    return EFG_OK;
  }
  EFG_CHECK_ARG(x && ws, "colsum: null pointer");
  ColsumPlan p = colsum_plan(rows, cols, row_stride, x);
  const bool rowwide = rowwide_plan(rows, cols, row_stride, x, &p);   // (the same summation tree as the ReLU-backward form)
  EFG_CHECK_ARG(ws_bytes >= sizeof(float) * (size_t)cols * (size_t)p.nblocks, "colsum: workspace too small");
  float* partial = p.nblocks == 1 ? out : static_cast<float*>(ws);
  unsigned* tickets = p.nblocks > 1 && p.nblocks <= kFusedBlocks ? ticket_slots(p.ygroups, st) : nullptr;
  const dim3 grid(p.nblocks, p.ygroups);
  if (rowwide)
    hipLaunchKernelGGL(colsum_rowwide_kernel<false>, grid, dim3(256), 0, st, x, (const float*)nullptr, (long long)rows,
                       (long long)row_stride, p.rows_per_block, cols, (float*)nullptr, partial, tickets, out);
  else if (p.vec)
    hipLaunchKernelGGL(colsum_partial_kernel<true>, grid, dim3(256), 0, st, x, (long long)rows, p.slots,
                       (long long)row_stride, p.rows_per_block, p.slot_log2, cols, partial, tickets, out);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<false>, grid, dim3(256), 0, st, x, (long long)rows, p.slots,
                       (long long)row_stride, p.rows_per_block, p.slot_log2, cols, partial, tickets, out);
  EFG_LAUNCH_CHECK();
  if (tickets) return EFG_OK;
  if (p.nblocks > 64) {
    hipLaunchKernelGGL(colsum_final_kernel<16>, dim3((unsigned)ceil_div(cols, 64)), dim3(1024), 0, st, partial,
                       p.nblocks, cols, out);
    EFG_LAUNCH_CHECK();
  } else if (p.nblocks > 1) {
    hipLaunchKernelGGL(colsum_final_kernel<4>, dim3((unsigned)ceil_div(cols, 64)), dim3(256), 0, st, partial,
                       p.nblocks, cols, out);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

// g_out[r][c] = y[r][c] > 0 ? g[r][c] : 0 and out[c] = sum_r g_out[r][c]; contiguous [rows, cols] matrices, cols % 4 == 0,
// 16-byte aligned; workspace as efg_colsum_workspace_bytes(rows, cols).  g_out may alias g.
extern "C" int efg_relu_bwd_colsum_f32(const float* g, const float* y, int64_t rows, int cols, float* g_out, float* out,
                                       void* ws, size_t ws_bytes, void* stream) {
  EFG_CHECK_ARG(rows >= 0 && cols >= 4 && cols % 4 == 0, "relu_bwd_colsum: bad shape %lld x %d", (long long)rows, cols);
  EFG_CHECK_ARG(out, "relu_bwd_colsum: null output");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    EFG_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * cols, st));
    return EFG_OK;
  }
  EFG_CHECK_ARG(g && y && g_out && ws, "relu_bwd_colsum: null pointer");
  EFG_CHECK_ARG(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(g_out)) & 15) == 0,
                "relu_bwd_colsum: operands must be 16-byte aligned");
  ColsumPlan p = colsum_plan(rows, cols, cols, g);
  const bool rowwide = rowwide_plan(rows, cols, cols, g, &p);
  EFG_CHECK_ARG(ws_bytes >= sizeof(float) * (size_t)cols * (size_t)p.nblocks, "relu_bwd_colsum: workspace too small");
  float* partial = p.nblocks == 1 ? out : static_cast<float*>(ws);
  unsigned* tickets = p.nblocks > 1 && p.nblocks <= kFusedBlocks ? ticket_slots(p.ygroups, st) : nullptr;
  if (rowwide)
    hipLaunchKernelGGL(colsum_rowwide_kernel<true>, dim3(p.nblocks, p.ygroups), dim3(256), 0, st, g, y, (long long)rows,
                       (long long)cols, p.rows_per_block, cols, g_out, partial, tickets, out);
  else
    hipLaunchKernelGGL(relu_bwd_colsum_partial_kernel, dim3(p.nblocks, p.ygroups), dim3(256), 0, st, g, y, (long long)rows, p.slots,
                       (long long)cols, p.rows_per_block, p.slot_log2, cols, g_out, partial, tickets, out);
  EFG_LAUNCH_CHECK();
  if (tickets) return EFG_OK;
  if (p.nblocks > 64) {
    hipLaunchKernelGGL(colsum_final_kernel<16>, dim3((unsigned)ceil_div(cols, 64)), dim3(1024), 0, st, partial, p.nblocks, cols,
                       out);
    EFG_LAUNCH_CHECK();
  } else if (p.nblocks > 1) {
    hipLaunchKernelGGL(colsum_final_kernel<4>, dim3((unsigned)ceil_div(cols, 64)), dim3(256), 0, st, partial, p.nblocks, cols, out);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

// Health check of the ticket ring of the CURRENT device (the trainer's periodic anomaly check).  Two symptoms are counted:
// (i) tickets drawn beyond a launch's block count (g_ticket_errors, seen by the kernels themselves), and (ii) ring words that
// are not zero AT REST: a counter that starts a launch at 0 < k < nblocks makes the (nblocks - k)th block sum rows that are
// not all written, resets the word, and the remaining k blocks leave it at k again -- no kernel can see that, so the whole
// ring (32 KB) is read back after a device synchronise, when no launch that draws tickets is in flight (the caller's
// contract: call it between steps), and every non-zero word is counted.  `reset` puts the ring and the counter back to rest.
extern "C" int efg_ticket_ring_errors(int64_t* count_out, int reset) {
  EFG_CHECK_ARG(count_out != nullptr, "ticket_ring_errors: null output");
  int dev = 0;
  EFG_HIP_TRY(hipGetDevice(&dev));
  EFG_HIP_TRY(hipDeviceSynchronize());   // (torch's streams are non-blocking: a copy on the null stream alone would not wait for them)
  unsigned v = 0;
  EFG_HIP_TRY(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ticket_errors), sizeof(v)));
  int64_t residue = 0;
  unsigned* ring = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_ring_mu);
    if (dev >= 0 && dev < 64) ring = g_ring[dev];
  }
  if (ring) {
    std::vector<unsigned> host(kRing);
    EFG_HIP_TRY(hipMemcpy(host.data(), ring, sizeof(unsigned) * kRing, hipMemcpyDeviceToHost));
    for (unsigned w : host) residue += w != 0;
    if (reset && residue) EFG_HIP_TRY(hipMemset(ring, 0, sizeof(unsigned) * kRing));
  }
  *count_out = (int64_t)v + residue;
  if (reset && v) {
    const unsigned zero = 0;
    EFG_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ticket_errors), &zero, sizeof(zero)));
  }
  return EFG_OK;
}
