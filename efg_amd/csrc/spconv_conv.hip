// Sparse convolution arithmetic for gfx950: output-stationary gathered implicit GEMM on
// v_mfma_f32_16x16x4_f32 (exact fp32, the dtype the reference trains in), forward / dgrad / wgrad.
//
// Replaces spconv's gather -> GEMM -> scatter-add per kernel offset as exercised by
// efg/modeling/backbones/sparse_net.py:79-98,120-165,273-309,485-545 (contract SURVEY.md B.6).
// No scatter and no atomics: with nbr[k][o] (spconv_index.hip) every output row is produced once,
//     out[o][:] = sum_k  in[nbr[k][o]][:] . W[:,k,:]^T
// and dgrad is the SAME kernel driven by the transposed table rnbr and transposed weights.
//
// forward / dgrad kernel, per wave (waves are independent; a 256-thread block is 4 of them):
//   * 16 output rows x all output channels: NT accumulator tiles of 16x16 (4 VGPRs each);
//   * the wave's nbr column block (kvol x 16 ints) is staged in LDS once; offsets k for which none
//     of the 16 rows has a neighbour are skipped entirely (rows are in spatial order, so validity is
//     spatially coherent);
//   * for an active k the 16 gathered input rows are read as contiguous 256-byte runs (one row per
//     load instruction, 16 loads in flight), staged in a wave-private LDS tile with row stride
//     CK+2 (bank-conflict-free ds_read_b32 of the A fragment), double-buffered against the MFMAs;
//   * the B fragments come straight from L1/L2 as 16-byte loads of weights pre-packed into MFMA
//     operand order (efg_spconv_pack_weight_f32): one load feeds four MFMAs.
// wgrad: per (row chunk, 64x64 block of dW, kernel offset) workgroup; valid pairs compacted into dense
// 64-pair tiles, G^T x gathered-X on the same MFMA, partials to a workspace, deterministic reduce.
#include "common.h"

#include <cstdlib>

namespace efg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCK = 64;           // channels per staged chunk
constexpr int kAStride = kCK + 2;  // 2 x odd -> conflict-free A-fragment reads
constexpr int kMaxKvol = 64;

__device__ __forceinline__ int round16(int x) { return (x + 15) & ~15; }

// ---- weight packing -----------------------------------------------------------------------
// packed[((k*R16 + r16)*NP + n)*16 + kk*4 + j] = W(red = r16*16 + 4*j + kk, n) for offset k,
// where (red, n) = (ci, co) forward, (co, ci) dgrad; zero padded to multiples of 16.
// for_dgrad bit 1 ("natural order", the 16-byte-gather tile kernel): red = r16*16 + 4*kk + j instead.
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ w, int cout, int kvol, int cin, int for_dgrad,
                                                 float* __restrict__ packed, unsigned bid, unsigned nblk) {
  const int natural = for_dgrad & 2, bf3 = for_dgrad & 4;
  for_dgrad &= 1;
  const int red = for_dgrad ? cout : cin, nn = for_dgrad ? cin : cout;
  if (bf3) {
    // split-precision arm of the tile kernel (spconv_tiles.hip, MODE & 4; red % 32 == 0, nn % 64 == 0): per (offset,
    // 32-channel step, n-tile of 16) 2 KB = the 64 lanes' 8 bf16 of W_hi, then of W_lo, lane = n % 16 + 16 * ((red % 32) / 8)
    typedef __bf16 bf16_t;
    bf16_t* out = reinterpret_cast<bf16_t*>(packed);
    const int c32n = red / 32, nt16 = nn / 16;
    const long long total = (long long)kvol * red * nn;
    for (long long e = (long long)bid * blockDim.x + threadIdx.x; e < total; e += (long long)nblk * blockDim.x) {
      const int n = (int)(e % nn);
      long long q = e / nn;
      const int r = (int)(q % red), k = (int)(q / red);
      const int co = for_dgrad ? r : n, ci = for_dgrad ? n : r;
      const float v = w[((long long)co * kvol + k) * cin + ci];
      const bf16_t hi = (bf16_t)v;
      const bf16_t lo = (bf16_t)(v - (float)hi);
      const long long idx = ((((long long)k * c32n + r / 32) * nt16 + n / 16) * 1024) + ((n % 16) + 16 * ((r % 32) / 8)) * 8 + (r % 8);
      out[idx] = hi;
      out[idx + 512] = lo;
    }
    return;
  }
  const int r16n = round16(red) / 16, np = round16(nn);
  const long long total = (long long)kvol * r16n * np * 16;
  // (+ the slack row kernels with NT > tiles read past the last n-tile: written as zeros here, no separate memset)
  for (long long e = (long long)bid * blockDim.x + threadIdx.x; e < total + 16 * 256;
       e += (long long)nblk * blockDim.x) {
    if (e >= total) {
      packed[e] = 0.0f;
      continue;
    }
    const int t = (int)(e & 15);
    const int kk = t >> 2, j = t & 3;
    long long q = e >> 4;
    const int n = (int)(q % np);
    q /= np;
    const int r16 = (int)(q % r16n);
    const int k = (int)(q / r16n);
    const int r = natural ? r16 * 16 + 4 * kk + j : r16 * 16 + 4 * j + kk;
    float v = 0.0f;
    if (r < red && n < nn) {
      const int co = for_dgrad ? r : n, ci = for_dgrad ? n : r;
      v = w[((long long)co * kvol + k) * cin + ci];
    }
    packed[e] = v;
  }
}

__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ w, int cout, int kvol, int cin,
                                                           int for_dgrad, float* __restrict__ packed) {
  pack_weight_body(w, cout, kvol, cin, for_dgrad, packed, blockIdx.x, gridDim.x);
}

// Every packed copy a model needs, in ONE launch (blockIdx.y = item): the layers' weights change together -- at the
// optimizer step -- so their MFMA-order copies are refreshed together instead of by two launches per convolution.
struct PackItem {
  const float* w;
  float* packed;
  int cout, kvol, cin, flags;
};
__global__ void __launch_bounds__(256) pack_weights_multi_kernel(const PackItem* __restrict__ items) {
  const PackItem it = items[blockIdx.y];
  pack_weight_body(it.w, it.cout, it.kvol, it.cin, it.flags, it.packed, blockIdx.x, gridDim.x);
}

// ---- forward / dgrad ------------------------------------------------------------------------
struct ConvArgs {
  const float* in;     // [m_in][cin]
  const float* wp;     // packed
  const float* bias;   // [cout] or null
  const int* nbr;      // [kvol][m_out]
  float* out;          // [m_out][cout]
  const int* order;    // null, or a permutation of the m_out rows: tile t computes rows order[16 t .. 16 t + 15]
  long long m_out;
  int cin, cout, kvol;
  int c16n;            // round16(cin)/16
  int np;              // round16(cout)
};

// KS: the kernel offsets of a 16-row tile are split over KS waves of the workgroup (split-K).  A wave walks its
// offsets one after the other -- gather, stash, MFMA -- so its run time is a serial chain of gather latencies, and a
// level only has a few waves per SIMD to overlap them (PMC: 34 % MFMA busy, 45 % of wave cycles waiting).  Splitting
// the offsets multiplies the waves and divides the chain; the partial accumulators meet in LDS.
template <int NT, int KV = 32, int KS = 1>
__global__ void __launch_bounds__(256) conv_fwd_kernel(ConvArgs a) {
  // wave-private staging: ONE A tile per wave (the wave itself orders compute -> refill; the prefetch
  // lives in registers) and the wave's rulebook block; 21 KB per workgroup for KV = 32
  __shared__ float a_tile[4][16 * kAStride];
  __shared__ int nbr_tile[4 / KS][KV * 16];  // one rulebook block per row tile (shared by its KS waves)
  __shared__ unsigned vmask_tile[4 / KS][KV];  // per offset: which of the 16 tile rows have a neighbour
  __shared__ int prow_tile[4 / KS][16];        // the tile's rows when a row order is given (else r0 + j)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int TILES = 4 / KS;          // row tiles per workgroup
  const int tile = wv / KS, part = wv % KS;
  // XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) by linear id;
  // rows are in canonical spatial order, so the 27 gathers of a row block hit rows that neighbouring row blocks
  // also read.  With the hardware order those neighbours sit on 8 different L2s (measured: 8x the algorithmic
  // bytes fetched past L2); renumbering gives every XCD one contiguous range of row blocks.
  unsigned bx = blockIdx.x, by = blockIdx.y;
  {
    const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x, total = gridDim.x * gridDim.y, per = total >> 3;
    if (per > 0 && lin < (per << 3)) {
      const unsigned nl = (lin & 7) * per + (lin >> 3);
      bx = nl % gridDim.x;
      by = nl / gridDim.x;
    }
  }
  const long long r0 = ((long long)bx * TILES + tile) * 16;
  const bool tile_ok = r0 < a.m_out;
  if (KS == 1 && !tile_ok) return;  // whole wave out of range (with KS == 1 waves never sync with each other)
  float* at0 = a_tile[wv];
  int* nb = nbr_tile[wv / KS];
  unsigned* vm = vmask_tile[wv / KS];
  const int n_tile0 = by * NT;  // first n-tile of this block

  // stage the wave's rulebook block and find the active offsets
  int* prow = prow_tile[wv / KS];
  if (a.order) {  // (uniform) rows of this tile through the caller's order: e.g. grouped by coordinate parity, so that
                  // the rows of a tile of a strided layer's dgrad need the same few offsets
    if (part == 0 && lane < 16) prow[lane] = (r0 + lane < a.m_out) ? a.order[r0 + lane] : -1;
    if (KS > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int e = lane + 64 * part; e < a.kvol * 16; e += 64 * KS) {
      const int k = e >> 4, j = e & 15;
      const int pr = prow[j];
      nb[e] = (pr >= 0) ? a.nbr[(long long)k * a.m_out + pr] : -1;
    }
  } else {
    for (int e = lane + 64 * part; e < a.kvol * 16; e += 64 * KS) {
      const int k = e >> 4, j = e & 15;
      nb[e] = (r0 + j < a.m_out) ? a.nbr[(long long)k * a.m_out + r0 + j] : -1;
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (KS > 1) __syncthreads();  // the tile's waves each staged a share of the block
  unsigned long long active = 0;
  {
    int seen = 0;
    for (int k = 0; k < a.kvol; ++k) {
      const int v = (lane < 16) ? nb[k * 16 + lane] : -1;
      const unsigned long long bal = __ballot(v >= 0);
      if (lane == 0) vm[k] = (unsigned)bal;  // the tile's waves all write the same word
      if (bal) {
        if (seen % KS == part) active |= 1ull << k;  // this wave's share of the tile's active offsets
        ++seen;
      }
    }
    if (!tile_ok) active = 0;
  }
  // The gathers only need byte offsets: turn the row numbers into offsets once (invalid -> row 0, a legal address;
  // its value is dropped through the scalar mask), so a gather costs one add + one select per row on the VALU.
  if (KS > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
  for (int e = lane + 64 * part; e < a.kvol * 16; e += 64 * KS) nb[e] = (int)((unsigned)max(nb[e], 0) * (unsigned)a.cin * 4u);
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (KS > 1) __syncthreads();

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float b = 0.0f;
    const int co = (n_tile0 + t) * 16 + (lane & 15);
    if (a.bias && co < a.cout && part == 0) b = a.bias[co];
    acc[t] = f32x4{b, b, b, b};
  }

  const int nchunk = (a.c16n * 16 + kCK - 1) / kCK;  // channel chunks of 64 per offset
  float pre[16];
  unsigned pre_mask = 0;

  // gather of (k, chunk) into registers: row j of the wave -> pre[j] (lane = channel)
  auto gather = [&](int k, int ch) {
    // Branch-free: a predicated load compiles to a branch + its own basic block, and the waitcnt pass then
    // serialises the 16 loads (one s_waitcnt vmcnt(0) per load).  Clamp the address, load always, select.
    // lanes past cin read a clamped (finite) channel: the packed weights are zero there, so nothing is selected
    // per lane; rows without a neighbour are dropped with the scalar mask of the offset
    const unsigned cc4 = (unsigned)min(ch * kCK + lane, a.cin - 1) * 4u;
    const unsigned vmk = (unsigned)__builtin_amdgcn_readfirstlane((int)vm[k]);
    unsigned offs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) offs[j] = (unsigned)nb[k * 16 + j] + cc4;
#pragma unroll
    for (int j = 0; j < 16; ++j) pre[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) + offs[j]);
    pre_mask = vmk;  // rows without a neighbour are zeroed at stash time (a select here would wait for the loads)
  };
  auto stash = [&](float* at) {
#pragma unroll
    for (int j = 0; j < 16; ++j) at[j * kAStride + lane] = ((pre_mask >> j) & 1u) ? pre[j] : 0.0f;
  };
  auto compute = [&](const float* at, int k, int ch) {
    const int c16_lo = ch * (kCK / 16);
    const int c16_hi = min(c16_lo + kCK / 16, a.c16n);
    const int m = lane & 15, kk = lane >> 4;
    for (int c16 = c16_lo; c16 < c16_hi; ++c16) {
      const float* ap = at + m * kAStride + (c16 - c16_lo) * 16 + kk;
      const float a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
      // 32-bit byte offset into the packed weights (a few MB): saddr + voffset loads
      const unsigned boff = ((((unsigned)k * (unsigned)a.c16n + (unsigned)c16) * (unsigned)a.np + (unsigned)(n_tile0 * 16 + m)) * 16u +
                             (unsigned)(kk * 4)) * 4u;
      float4 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        b[t] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.wp) + boff + (unsigned)t * 1024u);
      // k-step outer, n-tile inner: consecutive MFMAs hit different accumulators (a 16x16x4 f32 MFMA
      // issues every 32 cycles but its result is ready after 40)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b[t].w, acc[t], 0, 0, 0);
    }
  };

  // software pipeline over the (active offset, chunk) steps
  const int nsteps = __popcll(active) * nchunk;
  if (nsteps > 0) {
    unsigned long long rem = active;
    int k_cur = __ffsll((long long)rem) - 1, ch_cur = 0;
    gather(k_cur, ch_cur);
    stash(at0);
    for (int s = 0; s < nsteps; ++s) {
      // next step coordinates
      int k_nxt = k_cur, ch_nxt = ch_cur + 1;
      if (ch_nxt == nchunk) {
        ch_nxt = 0;
        rem &= rem - 1;
        k_nxt = rem ? __ffsll((long long)rem) - 1 : -1;
      }
      const bool more = (s + 1 < nsteps);
      if (more) gather(k_nxt, ch_nxt);  // global loads in flight during the MFMAs below
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      compute(at0, k_cur, ch_cur);
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all fragment reads done before the refill
      if (more) stash(at0);
      k_cur = k_nxt;
      ch_cur = ch_nxt;
    }
  }

  if (KS > 1) {
    // partial accumulators of the tile's other waves: through the (now idle) A tiles, 16 x 16 floats per n-tile
    __syncthreads();  // every wave is done with its A tile
    float* red = a_tile[wv];
    for (int t0 = 0; t0 < NT; t0 += 4) {  // 4 n-tiles (1024 floats) fit one A tile (16 x 66 floats)
      if (part != 0) {
#pragma unroll
        for (int t = t0; t < NT && t < t0 + 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((t - t0) * 4 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (part == 0) {
        for (int p = 1; p < KS; ++p) {
          const float* o = a_tile[wv + p];
#pragma unroll
          for (int t = t0; t < NT && t < t0 + 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += o[((t - t0) * 4 + r) * 64 + lane];
        }
      }
      __syncthreads();
    }
    if (part != 0 || !tile_ok) return;
  }
  // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int co = (n_tile0 + t) * 16 + (lane & 15);
    if (co < a.cout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = (lane >> 4) * 4 + r;
        const long long row = a.order ? (long long)prow[j] : ((r0 + j < a.m_out) ? r0 + j : -1);
        if (row >= 0) a.out[row * a.cout + co] = acc[t][r];
      }
    }
  }
}

// ---- wgrad ------------------------------------------------------------------------------------
// dW[co][k][ci] = sum_o G[o][co] * X[nbr[k][o]][ci].  One workgroup = (row chunk, 64x64 block of dW,
// ONE kernel offset k).  Only rows that really have a neighbour at k enter the MFMAs: the chunk's
// valid (out row, in row) pairs are compacted with ballot/popcount into an LDS queue and consumed as
// dense 64-pair tiles (strided convs have 1/8 of the rows valid per offset, SubM ~60 %).  Partial
// blocks go to a workspace; a second pass reduces them in a fixed order (deterministic).
struct WgradArgs {
  const float* in;    // [m_in][cin]
  const float* go;    // [m_out][cout]
  const int* nbr;     // [kvol][m_out]
  float* partial;     // [splits][kvol][cout][cin]
  long long m_out;
  int cin, cout, kvol;
  int rows_per_split;  // multiple of 256
  int nci_blk;         // ceil(cin / 64)
};

constexpr int kWStride = 64 + 16;  // row stride of the LDS tiles (== 16 mod 32: conflict-free fragment reads)
constexpr int kTP = 32;            // pairs per MFMA tile: 23 KB of LDS per workgroup -> 6 workgroups per CU
constexpr int kQueue = kTP + 256;

__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradArgs a) {
  __shared__ float g_tile[kTP * kWStride];   // kTP pairs x 64 co of grad_out
  __shared__ float x_tile[kTP * kWStride];   // kTP pairs x 64 ci of the gathered input
  __shared__ int q_out[kQueue], q_in[kQueue];
  __shared__ int wave_cnt[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // XCD-aware order: the kvol x (co, ci) workgroups of one row chunk re-read the same grad_out / input rows; deal
  // them to ONE XCD, consecutively, so those rows come from its L2 (hardware order: linear id round-robin over the
  // 8 XCDs, which spread a chunk's workgroups over all 8 L2s -- PMC: 10x the algorithmic bytes fetched past L2)
  int split = blockIdx.x, k = blockIdx.z, yb = blockIdx.y;
  if ((gridDim.x & 7) == 0) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned per = gridDim.y * gridDim.z, idx = lin >> 3, rest = idx % per;
    split = (int)((lin & 7) + 8 * (idx / per));
    yb = (int)(rest % gridDim.y);
    k = (int)(rest / gridDim.y);
  }
  const int co0 = (yb / a.nci_blk) * 64, ci0 = (yb % a.nci_blk) * 64;
  const long long row_lo = (long long)split * a.rows_per_split;
  const long long row_hi = min(row_lo + a.rows_per_split, a.m_out);
  const int m = lane & 15, kk = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One MFMA tile = kTP pairs.  load(): this wave's share of the tile's rows into registers (branch-free, clamped
  // address + select); stash(): registers -> LDS; compute(): the MFMAs.  Tiles that are already in the queue are
  // loaded while the previous tile computes.
  constexpr int JW = kTP / 4;  // tile rows per wave
  float gr[JW], xr[JW];
  unsigned okbits = 0;  // validity of the loaded rows (the selects wait for the data: they run at stash time)
  const int gco = min(co0 + lane, a.cout - 1), gci = min(ci0 + lane, a.cin - 1);
  const bool co_ok = co0 + lane < a.cout, ci_ok = ci0 + lane < a.cin;
  auto load = [&](int head) {
    okbits = 0;
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      const int j = wv + 4 * jj;
      const int o = q_out[head + j], i = q_in[head + j];
      // 32-bit byte offsets (saddr + voffset loads; wgrad checks both tensors are below 4 GB)
      const float g = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.go) +
                                                      ((unsigned)max(o, 0) * (unsigned)a.cout + (unsigned)gco) * 4u);
      const float x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) +
                                                      ((unsigned)max(i, 0) * (unsigned)a.cin + (unsigned)gci) * 4u);
      gr[jj] = g;
      xr[jj] = x;
      okbits |= (o >= 0 ? 1u : 0u) << jj;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      const bool ok = (okbits >> jj) & 1u;
      g_tile[(wv + 4 * jj) * kWStride + lane] = (ok && co_ok) ? gr[jj] : 0.f;
      x_tile[(wv + 4 * jj) * kWStride + lane] = (ok && ci_ok) ? xr[jj] : 0.f;
    }
  };
  auto compute = [&]() {
    // wave wv owns the co tile co0 + wv*16 and the 4 ci tiles; reduction dim = pairs, 4 per MFMA
#pragma unroll 4
    for (int s = 0; s < kTP / 4; ++s) {
      const int row = s * 4 + kk;
      const float av = g_tile[row * kWStride + wv * 16 + m];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float bv = x_tile[row * kWStride + t * 16 + m];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  };
  // consume every full tile of the queue from `head` on; returns the new head
  auto drain = [&](int head, int qn) {
    if (qn - head < kTP) return head;
    __syncthreads();  // queue entries visible
    load(head);
    while (true) {
      __syncthreads();  // previous tile fully consumed
      stash();
      __syncthreads();
      head += kTP;
      const bool more = qn - head >= kTP;
      if (more) load(head);  // in flight during the MFMAs below
      compute();
      if (!more) break;
    }
    return head;
  };

  int qn = 0;  // block-uniform queue length
  // the table column of the NEXT 256 rows is requested before this segment's pairs are multiplied (its round trip used
  // to sit between two drains)
  int r_nxt = (row_lo + threadIdx.x < row_hi) ? a.nbr[(long long)k * a.m_out + row_lo + threadIdx.x] : -1;
  for (long long seg = row_lo; seg < row_hi; seg += 256) {
    const long long row = seg + threadIdx.x;
    const int r = r_nxt;
    r_nxt = (row + 256 < row_hi) ? a.nbr[(long long)k * a.m_out + row + 256] : -1;
    const unsigned long long mask = __ballot(r >= 0);
    if (lane == 0) wave_cnt[wv] = __popcll(mask);
    __syncthreads();
    int base = qn;
    for (int w = 0; w < wv; ++w) base += wave_cnt[w];
    if (r >= 0) {
      const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      q_out[pos] = (int)row;
      q_in[pos] = r;
    }
    qn += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    int head = 0;
    head = drain(head, qn);
    __syncthreads();  // all reads of the queue / wave_cnt done
    if (head > 0) {   // move the remainder (< kTP pairs) to the front
      const int rem = qn - head;
      int to = -1, ti = -1;
      if ((int)threadIdx.x < rem) {
        to = q_out[head + threadIdx.x];
        ti = q_in[head + threadIdx.x];
      }
      __syncthreads();
      if ((int)threadIdx.x < rem) {
        q_out[threadIdx.x] = to;
        q_in[threadIdx.x] = ti;
      }
      qn = rem;
    }
  }
  if (qn > 0) {
    __syncthreads();
    if ((int)threadIdx.x >= qn && threadIdx.x < kTP) {
      q_out[threadIdx.x] = -1;
      q_in[threadIdx.x] = -1;
    }
    drain(0, kTP);
  }
  // partial block: row (co) = (lane>>4)*4 + reg, col (ci) = lane & 15
  float* p = a.partial + ((long long)split * a.kvol + k) * a.cout * a.cin;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ci = ci0 + t * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wv * 16 + (lane >> 4) * 4 + r;
      if (co < a.cout && ci < a.cin) p[(long long)co * a.cin + ci] = acc[t][r];
    }
  }
}

// ---- wgrad for the low-channel layers (cin <= 16, cout <= 32: the stem, 5 -> 16 -> 16 -> 32) -------------------
// The 64 x 64 block kernel above spends most of its MFMA tile on padding there and launches kvol workgroups per row
// chunk that each compact and re-read the same rows: 226 us for 0.9 GFLOP on the 117 697-site input level.  These
// layers are not arithmetic at all (a few GFLOP against ~35 MB of rows and table), so: one workgroup = one row chunk
// x a group of <= 4 kernel offsets; the wave walks its 256 rows four at a time, loads the grad_out fragment of the
// four rows ONCE and feeds it to one 16x16x4 MFMA per (offset, 16 output channels) with the gathered input row as the
// other operand -- no compaction, rows without a neighbour contribute a zero operand.  dW[k] lives in registers
// (<= 4 offsets x 2 tiles x 4); the four waves are summed through LDS in a fixed order and written as one partial
// block per row chunk, which the same wgrad_reduce_kernel folds.

template <int T, int KJ>  // T = 16-channel tiles of cout, KJ = offsets per group (compile-time bound of the loops)
__global__ void __launch_bounds__(256) conv_wgrad_small_kernel(WgradArgs a) {
  __shared__ float red[3][KJ * T * 4 * 64];  // waves 1..3 park their accumulators here
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;  // A: ci = m of row kk;  B: co = 16 t + m of row kk
  const int g = blockIdx.y, kg = gridDim.y;  // offset group: k = g, g + kg, g + 2 kg, ...
  const long long chunk_lo = (long long)blockIdx.x * a.rows_per_split;
  const long long chunk_hi = min(chunk_lo + a.rows_per_split, a.m_out);
  const int nk = (a.kvol - g + kg - 1) / kg;  // offsets of this group (<= KJ)
  f32x4 acc[KJ][T];
#pragma unroll
  for (int j = 0; j < KJ; ++j)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool ci_ok = m < a.cin;
  const int ci = ci_ok ? m : 0;
  // rows of this wave: steps of 4 rows, the 4 waves interleaved; kU steps are loaded together (table entries first,
  // then the rows they point to, then the MFMAs) so that each wave keeps kU * (KJ + T) independent loads in flight --
  // with two waves per SIMD the memory latency is all there is to hide
  constexpr int kU = 4;
  for (long long r = chunk_lo + wv * 4; r < chunk_hi; r += 16 * kU) {
    bool row_ok[kU];
    long long rr[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const long long row = r + 16 * u + kk;
      row_ok[u] = row < chunk_hi;
      rr[u] = row_ok[u] ? row : chunk_lo;
    }
    int idx[kU][KJ];
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int j = 0; j < KJ; ++j) idx[u][j] = (j < nk) ? a.nbr[(long long)(g + kg * j) * a.m_out + rr[u]] : -1;
    float bfrag[kU][T];
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int co = 16 * t + m;
        const float v = a.go[rr[u] * a.cout + (co < a.cout ? co : 0)];
        bfrag[u][t] = (row_ok[u] && co < a.cout) ? v : 0.f;
      }
    float afrag[kU][KJ];
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int j = 0; j < KJ; ++j) afrag[u][j] = a.in[(long long)max(idx[u][j], 0) * a.cin + ci];
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        const float av = (idx[u][j] >= 0 && row_ok[u] && ci_ok) ? afrag[u][j] : 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t)
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag[u][t], acc[j][t], 0, 0, 0);
      }
  }
  // waves 1..3 -> LDS, wave 0 sums in wave order and writes partial[split][k][co][ci]
  if (wv > 0) {
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wv - 1][((j * T + t) * 4 + q) * 64 + lane] = acc[j][t][q];
  }
  __syncthreads();
  if (wv == 0) {
    // D layout of 16x16x4: lane holds C[row = 4 * (lane >> 4) + q][col = lane & 15] = dW[ci = 4 kk + q][co = 16 t + m]
    float* out = a.partial + (long long)blockIdx.x * a.kvol * a.cout * a.cin;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      if (j >= nk) break;
      const int k = g + kg * j;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int co = 16 * t + m;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cix = 4 * kk + q;
          float v = acc[j][t][q];
#pragma unroll
          for (int w = 0; w < 3; ++w) v += red[w][((j * T + t) * 4 + q) * 64 + lane];
          if (co < a.cout && cix < a.cin) out[((long long)k * a.cout + co) * a.cin + cix] = v;
        }
      }
    }
  }
}

// grad_w[co][k][ci] = sum_s partial[s][k][co][ci].  64 elements per workgroup (coalesced), the 4 waves take every
// fourth split with 4 loads in flight each, then a fixed-order sum through LDS: with ~120 splits on the low-channel
// layers a one-thread-per-element loop over the splits was a chain of 120 dependent-latency loads (40-60 us).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int kvol,
                                                            int cout, int cin, float* __restrict__ gw) {
  __shared__ float sm[4][64];
  const long long per = (long long)kvol * cout * cin;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < per) {
    int sp = wv;
    for (; sp + 12 < splits; sp += 16) {
      s0 += partial[(long long)sp * per + e];
      s1 += partial[(long long)(sp + 4) * per + e];
      s2 += partial[(long long)(sp + 8) * per + e];
      s3 += partial[(long long)(sp + 12) * per + e];
    }
    for (; sp < splits; sp += 4) s0 += partial[(long long)sp * per + e];
  }
  sm[wv][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wv == 0 && e < per) {
    const float s = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
    const int ci = (int)(e % cin);
    const long long q = e / cin;
    const int co = (int)(q % cout);
    const int k = (int)(q / cout);
    gw[((long long)co * kvol + k) * cin + ci] = s;
  }
}

struct WgradPlan {
  int splits, rows_per_split, nco_blk, nci_blk;
  size_t bytes;
};

WgradPlan wgrad_plan(int64_t m_out, int cin, int cout, int kvol) {
  WgradPlan p;
  p.nco_blk = (cout + 63) / 64;
  p.nci_blk = (cin + 63) / 64;
  const size_t per = (size_t)kvol * cout * cin * 4;
  // ~2048 workgroups (8 per CU) when the level is large enough (as fast as 4096 with half the partial-block traffic;
  // 1024 is 15-30 % slower: scripts/tile_sweep.sh), >= 512 rows per chunk, <= 128 MB partials
  constexpr int64_t fill_env = 2048;
  const int64_t by_fill = std::max<int64_t>(1, fill_env / ((int64_t)p.nco_blk * p.nci_blk * kvol));
  const int64_t by_rows = std::max<int64_t>(1, ceil_div(std::max<int64_t>(m_out, 1), 512));
  const int64_t by_mem = std::max<int64_t>(1, (int64_t)((128ull << 20) / std::max<size_t>(per, 1)));
  int64_t s = std::min(std::min(by_fill, by_rows), by_mem);
  int64_t rows = ceil_div(std::max<int64_t>(m_out, 1), s);
  rows = ceil_div(rows, 256) * 256;
  s = std::max<int64_t>(1, ceil_div(std::max<int64_t>(m_out, 1), rows));
  if (s >= 8) s = ceil_div(s, 8) * 8;  // multiple of 8: lets the kernel give every XCD whole row chunks (empty ones cost nothing)
  p.splits = (int)s;
  p.rows_per_split = (int)rows;
  p.bytes = (size_t)p.splits * per;
  return p;
}

template <int NT>
void launch_fwd(const ConvArgs& a, int nblk_y, int ks, hipStream_t stream) {
  if (a.kvol <= 32 && ks == 2) {
    hipLaunchKernelGGL((conv_fwd_kernel<NT, 32, 2>), dim3((unsigned)ceil_div(a.m_out, 32), nblk_y), dim3(256), 0, stream, a);
    return;
  }
  if (a.kvol <= 32 && ks == 4) {
    hipLaunchKernelGGL((conv_fwd_kernel<NT, 32, 4>), dim3((unsigned)ceil_div(a.m_out, 16), nblk_y), dim3(256), 0, stream, a);
    return;
  }
  const unsigned gx = (unsigned)ceil_div(a.m_out, 64);
  if (a.kvol <= 32) hipLaunchKernelGGL((conv_fwd_kernel<NT, 32>), dim3(gx, nblk_y), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_fwd_kernel<NT, kMaxKvol>), dim3(gx, nblk_y), dim3(256), 0, stream, a);
}

int run_conv(const float* in, int64_t m_in, int cin, const float* wp, const float* bias, int cout, int kvol,
             const int* nbr, int64_t m_out, float* out, hipStream_t stream, const int* order = nullptr) {
  EFG_CHECK_ARG(cin >= 1 && cout >= 1, "spconv: bad channel counts");
  EFG_CHECK_ARG(kvol >= 1 && kvol <= kMaxKvol, "spconv: kernel volume must be in [1,%d], got %d", kMaxKvol, kvol);
  if (m_out == 0) return EFG_OK;
  EFG_CHECK_ARG(m_in >= 0 && (unsigned long long)m_in * (unsigned long long)cin * 4ull < (1ull << 32),
                "spconv: input features of %lld x %d floats exceed the 4 GB the gather addresses", (long long)m_in, cin);
  ConvArgs a;
  a.in = in;
  a.wp = wp;
  a.bias = bias;
  a.nbr = nbr;
  a.out = out;
  a.order = order;
  a.m_out = m_out;
  a.cin = cin;
  a.cout = cout;
  a.kvol = kvol;
  a.c16n = (cin + 15) / 16;
  a.np = (cout + 15) / 16 * 16;
  const int ntiles = a.np / 16;
  // n-tiles per wave (NT): as many as possible (A-tile reuse) while the launch still has >= ~2 waves
  // per SIMD on the whole chip; small levels (a few thousand rows x 256 channels) otherwise run one
  // long serial MFMA chain per wave on a third of the CUs.  Wider outputs tile over grid.y.
  const long long row_waves = ceil_div(m_out, 16);
  int nt = 16;
  // With the kernel offsets split over the 4 waves of a workgroup (KS, below) a row tile already is 4 waves, so the
  // 3x3x3 layers fill the chip with fewer tiles: 1400 instead of 2048 lets the 256-channel layers (371 row tiles) take
  // 4 n-tiles per wave instead of 2 -- half the gathers per MFMA: 205 -> 170 us, 60 -> 72 TFLOP/s (EFG_CONV_FILL).
  constexpr long long fill_env = 0;
  constexpr int ks_env = 4;
  const long long fill = fill_env > 0 ? fill_env : ((kvol >= 8 && ks_env > 1) ? 1400 : 2048);
  while (nt > 1 && (nt / 2 >= ntiles || row_waves * ((ntiles + nt - 1) / nt) < fill)) nt >>= 1;
  if (nt > ntiles) nt = ntiles >= 16 ? 16 : ntiles >= 8 ? 8 : ntiles >= 4 ? 4 : ntiles >= 2 ? 2 : 1;
  while (nt < ntiles && nt < 16 && (ntiles % nt) != 0) nt <<= 1;  // keep grid.y exact where possible
  const int ny = (ntiles + nt - 1) / nt;
  // split-K over the 4 waves of a workgroup for the 3x3x3 kernels with 2-4 n-tiles per wave (measured -10 %);
  // 16-channel outputs (NT = 1) and the 3-offset z-collapsing heads gain nothing.  EFG_CONV_KS overrides (1 | 2 | 4).
  const int ks = (kvol >= 8) ? ks_env : 1;
  switch (nt) {
    case 16: launch_fwd<16>(a, ny, 1, stream); break;
    case 8: launch_fwd<8>(a, ny, 1, stream); break;
    case 4: launch_fwd<4>(a, ny, ks, stream); break;
    case 2: launch_fwd<2>(a, ny, ks, stream); break;
    default: launch_fwd<1>(a, ny, 1, stream); break;
  }
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" size_t efg_spconv_packed_weight_bytes(int cout, int kvol, int cin, int for_dgrad) {
  if (cout < 1 || cin < 1 || kvol < 1) return 0;
  for_dgrad &= 1;  // (bit 1 selects the natural channel order, same size)
  const int red = for_dgrad ? cout : cin, nn = for_dgrad ? cin : cout;
  // + one tile row of slack: kernels with NT > tiles read (zero-weight) past the last n-tile
  return ((size_t)kvol * ((red + 15) / 16) * ((nn + 15) / 16 * 16) * 16 + 16 * 256) * sizeof(float);
}

extern "C" int efg_spconv_pack_weight_f32(const float* weight, int cout, int kvol, int cin, int for_dgrad,
                                          float* packed, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(cout >= 1 && cin >= 1 && kvol >= 1, "spconv: bad weight shape");
  if (for_dgrad & 4) {   // split-precision layout of the tile kernel's A/B arm: whole 32-channel steps, whole n-tile groups
    const int red = (for_dgrad & 1) ? cout : cin, nn = (for_dgrad & 1) ? cin : cout;
    EFG_CHECK_ARG(!(for_dgrad & 2) && red % 32 == 0 && nn % 64 == 0,
                  "spconv: the bf16x3 weight layout needs red %% 32 == 0 and n %% 64 == 0 (got %d, %d)", red, nn);
  }
  const size_t bytes = efg_spconv_packed_weight_bytes(cout, kvol, cin, for_dgrad);
  const long long total = (long long)(bytes / sizeof(float));
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256),
                     0, stream, weight, cout, kvol, cin, for_dgrad, packed);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_spconv_pack_weights_multi(const void* items_dev, int n, void* stream) {
  EFG_CHECK_ARG(n >= 0 && (n == 0 || items_dev), "pack_weights_multi: bad arguments");
  static_assert(sizeof(PackItem) == 32, "PackItem is 2 pointers + 4 ints (the Python side writes it as 4 int64)");
  if (n == 0) return EFG_OK;
  // (256 workgroups per item: a thread of the largest layer -- 256 x 27 x 256 weights -- then walks 27 elements, one dependent
  // load each, instead of 108: the launch was 89 us of load latency)
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(256, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const PackItem*>(items_dev));
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_spconv_forward_f32(const float* in_feat, int64_t m_in, int cin, const float* packed_weight,
                                      const float* bias, int cout, int kvol, const int32_t* nbr, int64_t m_out,
                                      float* out_feat, void* stream) {
  return run_conv(in_feat, m_in, cin, packed_weight, bias, cout, kvol, nbr, m_out, out_feat, (hipStream_t)stream);
}

extern "C" int efg_spconv_dgrad_f32(const float* grad_out, int64_t m_out, int cout, const float* packed_weight,
                                    int cin, int kvol, const int32_t* rnbr, int64_t m_in, const int32_t* row_order,
                                    float* grad_in, void* stream) {
  return run_conv(grad_out, m_out, cout, packed_weight, nullptr, cin, kvol, rnbr, m_in, grad_in, (hipStream_t)stream,
                  row_order);
}

// ---- row order by coordinate parity ---------------------------------------------------------------------------
// The dgrad of a stride-2 convolution computes one row per FINE site, and which of the 27 offsets reach a parent
// depends only on the parity of the site's (z, y, x): 8 classes with ~3.4 offsets each.  Neighbouring rows of the
// canonical order differ in parity, so a 16-row tile runs ~10 offsets (scripts/ubench/tile_waste.py: x2.9).  This
// orders the rows by class (within a class in the order the waves arrive, i.e. roughly spatially); results do not
// depend on the order -- a row's sum runs over its own offsets in ascending k whatever its tile mates are.
namespace efg {
namespace {
__device__ __forceinline__ int parity_class(const int* __restrict__ idx, long long r) {
  const int4 v = *reinterpret_cast<const int4*>(idx + r * 4);  // (b, z, y, x)
  return ((v.y & 1) << 2) | ((v.z & 1) << 1) | (v.w & 1);
}

constexpr int kParRows = 1024;  // rows per workgroup (4 per thread)

// class totals: ballots -> LDS counters -> 8 global atomics per workgroup (one atomic per wave and class on 8 shared
// addresses took 27 us for 117k rows)
__global__ void __launch_bounds__(256) parity_count_kernel(const int* __restrict__ idx, long long m, int* __restrict__ counts) {
  __shared__ int sc[8];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 8) sc[threadIdx.x] = 0;
  __syncthreads();
  const long long r0 = (long long)blockIdx.x * kParRows;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long r = r0 + j * 256 + threadIdx.x;
    const int c = r < m ? parity_class(idx, r) : -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long bal = __ballot(c == k);
      if (lane == 0 && bal) atomicAdd(&sc[k], __popcll(bal));
    }
  }
  __syncthreads();
  if (threadIdx.x < 8 && sc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sc[threadIdx.x]);
}

// a workgroup reserves one range per class (8 global atomics), its waves take sub-ranges from LDS cursors
__global__ void __launch_bounds__(256) parity_scatter_kernel(const int* __restrict__ idx, long long m, const int* __restrict__ counts,
                                                              int* __restrict__ cursor, int* __restrict__ order) {
  __shared__ int sc[8], sbase[8], scur[8];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 8) sc[threadIdx.x] = 0;
  __syncthreads();
  const long long r0 = (long long)blockIdx.x * kParRows;
  int cls[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long r = r0 + j * 256 + threadIdx.x;
    cls[j] = r < m ? parity_class(idx, r) : -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long bal = __ballot(cls[j] == k);
      if (lane == 0 && bal) atomicAdd(&sc[k], __popcll(bal));
    }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    int before = 0;
    for (int k = 0; k < (int)threadIdx.x; ++k) before += counts[k];
    sbase[threadIdx.x] = before + (sc[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], sc[threadIdx.x]) : 0);
    scur[threadIdx.x] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long r = r0 + j * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long bal = __ballot(cls[j] == k);
      if (bal) {
        int start = 0;
        if (lane == 0) start = atomicAdd(&scur[k], __popcll(bal));
        start = __shfl(start, 0, 64);
        if (cls[j] == k) order[sbase[k] + start + __popcll(bal & ((1ull << lane) - 1ull))] = (int)r;
      }
    }
  }
}
}  // namespace
}  // namespace efg

extern "C" int efg_spconv_parity_order(const int32_t* indices, int64_t m, int32_t* order, void* ws, size_t ws_bytes,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(m >= 0 && m < (1ll << 31), "parity_order: bad row count");
  if (m == 0) return EFG_OK;
  EFG_CHECK_ARG(indices && order && ws && ws_bytes >= 64, "parity_order: null pointer / workspace < 64 bytes");
  int* counts = static_cast<int*>(ws);
  EFG_HIP_TRY(hipMemsetAsync(counts, 0, 64, stream));
  const int blocks = (int)ceil_div(m, kParRows);
  hipLaunchKernelGGL(parity_count_kernel, dim3(blocks), dim3(256), 0, stream, indices, (long long)m, counts);
  hipLaunchKernelGGL(parity_scatter_kernel, dim3(blocks), dim3(256), 0, stream, indices, (long long)m, counts, counts + 8,
                     order);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" size_t efg_spconv_wgrad_workspace_bytes(int64_t m_out, int cin, int cout, int kvol) {
  if (cin < 1 || cout < 1 || kvol < 1 || m_out < 0) return 0;
  return wgrad_plan(m_out, cin, cout, kvol).bytes + 256;
}

extern "C" int efg_spconv_wgrad_f32(const float* in_feat, int64_t m_in, int cin, const float* grad_out,
                                    int64_t m_out, int cout, int kvol, const int32_t* nbr, float* grad_w, void* ws,
                                    size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(cin >= 1 && cout >= 1 && kvol >= 1 && kvol <= kMaxKvol, "spconv wgrad: bad sizes");
  EFG_CHECK_ARG(m_in >= 0 && (unsigned long long)m_in * cin * 4ull < (1ull << 32) &&
                    (unsigned long long)(m_out > 0 ? m_out : 0) * cout * 4ull < (1ull << 32),
                "spconv wgrad: feature tensors must be smaller than 4 GB");
  if (m_out == 0) {
    EFG_HIP_TRY(hipMemsetAsync(grad_w, 0, (size_t)cout * kvol * cin * 4, stream));
    return EFG_OK;
  }
  const WgradPlan p = wgrad_plan(m_out, cin, cout, kvol);
  if (ws_bytes < p.bytes) {
    set_error("spconv wgrad workspace too small: need %zu bytes, got %zu", p.bytes + 256, ws_bytes);
    return EFG_E_WORKSPACE;
  }
  WgradArgs a;
  a.in = in_feat;
  a.go = grad_out;
  a.nbr = nbr;
  a.partial = static_cast<float*>(ws);
  a.m_out = m_out;
  a.cin = cin;
  a.cout = cout;
  a.kvol = kvol;
  a.rows_per_split = p.rows_per_split;
  a.nci_blk = p.nci_blk;
  constexpr bool small_env = true;
  if (small_env && cin <= 16 && cout <= 32 && m_in == m_out) {
    // low-channel submanifold layers (dense tables: ~15 of 27 offsets per row; the strided stem conv has 4 of 27 and
    // keeps the compacting kernel): one workgroup per (row chunk, group of <= 4 offsets), see conv_wgrad_small_kernel
    const dim3 grid(p.splits, (unsigned)ceil_div(kvol, 4));
    if (cout <= 16)
      hipLaunchKernelGGL((conv_wgrad_small_kernel<1, 4>), grid, dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((conv_wgrad_small_kernel<2, 4>), grid, dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(p.splits, p.nco_blk * p.nci_blk, kvol), dim3(256), 0, stream, a);
  }
  EFG_LAUNCH_CHECK();
  const long long per = (long long)kvol * cout * cin;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(per, 64)), dim3(256), 0, stream, a.partial, p.splits,
                     kvol, cout, cin, grad_w);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
