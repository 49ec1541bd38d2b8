// Sparse convolution, workgroup-cooperative variant for the large levels: TM row tiles x TN groups of 64 output
// channels per workgroup, all TM*TN waves walking the same (offset, 64-channel chunk) steps in lockstep.
//
// Why: in conv_tile_kernel every wave fetches its own weight fragments from the vector L1 -- 256 bytes per MFMA,
// ~50 % of the L1's peak bandwidth at the matrix pipe's rate (PMC, profiles/r02a_pmc_*) -- and gathers its own copy
// of the A rows for each 64-channel output group.  Here the weights of a step (16 KB per channel group) are copied
// into LDS ONCE per workgroup and read by its TM waves with conflict-free ds_read_b128 (weights are packed in the
// order the lanes consume them), and a row tile's gathered A chunk is shared by its TN waves:
//     bytes through L1 per MFMA = 64 / TN (A) + 256 / TM (B)        (TM = 8, TN = 1: 96 instead of 320).
// Global loads for step s+1 are issued before the MFMAs of step s and land in registers; they go to the other LDS
// buffer after the MFMAs -- one barrier per step, every load has a full step (>= 2048 matrix-pipe cycles) to arrive.
// Rows come from the tile plan of spconv_tiles.hip (mask-sorted: the tiles of a workgroup need nearly the same
// offsets; a wave whose tile lacks the current offset idles that step, the loop runs over the union).
#include "common.h"

#include <cstdlib>

namespace efg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCKw = 64;           // channels per step
constexpr int kAStrW = kCKw + 2;   // A tile row stride (floats)
constexpr int kATile = 16 * kAStrW;  // floats per A tile
constexpr int kBGroup = 4 * 4 * 256;  // floats of one channel group's step: [4 c16][4 n-tiles][64 lanes][4]

// ---- weights in lane order --------------------------------------------------------------------------------------
// packed[((((k * C16 + c16) * NTT + t) * 64 + lane) * 4 + j] = W(red = c16*16 + 4*j + (lane >> 4), n = t*16 + (lane & 15))
// (red, n) = (ci, co) forward, (co, ci) dgrad; NTT = n-tiles rounded up to a multiple of 4; zero padded.
__global__ void __launch_bounds__(256) pack_weight_lanes_kernel(const float* __restrict__ w, int cout, int kvol, int cin,
                                                                 int for_dgrad, float* __restrict__ packed) {
  const int red = for_dgrad ? cout : cin, nn = for_dgrad ? cin : cout;
  const int c16n = (red + 15) / 16, ntt = ((nn + 15) / 16 + 3) / 4 * 4;
  const long long total = (long long)kvol * c16n * ntt * 256;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 3), lane = (int)((e >> 2) & 63);
    long long q = e >> 8;
    const int t = (int)(q % ntt);
    q /= ntt;
    const int c16 = (int)(q % c16n);
    const int k = (int)(q / c16n);
    const int r = c16 * 16 + 4 * j + (lane >> 4), n = t * 16 + (lane & 15);
    float v = 0.0f;
    if (r < red && n < nn) {
      const int co = for_dgrad ? r : n, ci = for_dgrad ? n : r;
      v = w[((long long)co * kvol + k) * cin + ci];
    }
    packed[e] = v;
  }
}

struct WgArgs {
  const float* in;
  const float* wp;      // lane-order packed weights
  const float* bias;
  const int* rows;      // tile plan (spconv_tiles.hip)
  const int* nb;
  const unsigned* vm;
  float* out;
  long long n_tiles;
  int cin, cout, kvol, c16n, ntt;
  int flip;
};

template <int TM, int TN>
__global__ void __launch_bounds__(TM * TN * 64) conv_wg_kernel(WgArgs a) {
  constexpr int W = TM * TN, NT = 4;
  constexpr int RJ = 16 / TN;                      // A rows gathered per wave and step
  constexpr int BV = TN * 1024 / (W * 64);         // float4 of B copied per thread and step (= 16 / TM)
  extern __shared__ float lds[];
  float* ldsA = lds;                               // [2][TM][kATile]
  float* ldsB = ldsA + 2 * TM * kATile;            // [2][TN][kBGroup]
  int* nbs = reinterpret_cast<int*>(ldsB + 2 * TN * kBGroup);  // [TM][512] byte offsets [col][16]
  unsigned* s_cols = reinterpret_cast<unsigned*>(nbs + TM * 512);  // [TM]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tm = wv / TN, tn = wv % TN;
  // XCD-aware order: consecutive workgroups (neighbouring sorted tiles read the same input rows) on one XCD
  unsigned bx = blockIdx.x, by = blockIdx.y;
  {
    const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x, total = gridDim.x * gridDim.y, per = total >> 3;
    if (per > 0 && lin < (per << 3)) {
      const unsigned nl = (lin & 7) * per + (lin >> 3);
      bx = nl % gridDim.x;
      by = nl / gridDim.x;
    }
  }
  const long long t = (long long)bx * TM + tm;     // this wave's 16-row tile
  const bool tile_ok = t < a.n_tiles;
  const int ngrp0 = by * TN;                       // first 64-channel output group of the workgroup
  const int n_tile0 = (ngrp0 + tn) * NT;           // first 16-column tile of this wave

  // ---- plan of the tile: masks in registers, neighbour block as byte offsets in LDS --------------------------
  const unsigned vmr = (tile_ok && lane < 32) ? a.vm[t * 32 + lane] : 0u;
  const int prow = (tile_ok && lane < 16) ? a.rows[t * 16 + lane] : -1;
  if (tn == 0) {
    if (lane == 0) s_cols[tm] = (unsigned)__builtin_amdgcn_readlane((int)vmr, 31);
    if (tile_ok) {
      const int* src = a.nb + t * a.kvol * 16;
      for (int e = lane; e < a.kvol * 16; e += 64) nbs[tm * 512 + e] = (int)((unsigned)max(src[e], 0) * (unsigned)a.cin * 4u);
    }
  }
  __syncthreads();
  unsigned cols = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) cols |= s_cols[i];
  cols = (unsigned)__builtin_amdgcn_readfirstlane((int)cols);

  f32x4 acc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    float b = 0.0f;
    const int co = (n_tile0 + q) * 16 + (lane & 15);
    if (a.bias && co < a.cout) b = a.bias[co];
    acc[q] = f32x4{b, b, b, b};
  }
  const int nchunk = (a.c16n * 16 + kCKw - 1) / kCKw;
  const int total_steps = __popc(cols) * nchunk;
  if (total_steps == 0) return;  // (uniform over the workgroup)

  float pre[RJ];
  unsigned pre_m = 0;   // rows of the tile with a neighbour at the loaded column (0: nothing loaded)
  float4 bre[BV];

  // loads of one step into registers: this wave's share of its tile's A rows + this thread's share of the weights
  auto load_step = [&](int col, int ch) {
    pre_m = (unsigned)__builtin_amdgcn_readlane((int)vmr, col);
    if (pre_m) {
      const unsigned cc4 = (unsigned)min(ch * kCKw + lane, a.cin - 1) * 4u;
#pragma unroll
      for (int j = 0; j < RJ; ++j)
        pre[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) +
                                                 ((unsigned)nbs[tm * 512 + col * 16 + tn * RJ + j] + cc4));
    }
    const int k = a.flip ? (a.kvol - 1 - col) : col;
#pragma unroll
    for (int v = 0; v < BV; ++v) {
      const int e = tid + v * (W * 64);            // float4 index in the step's [TN][4 c16][4 t][64 lanes] block
      const int g = e >> 10, r = e & 1023, c16i = r >> 8, q = r & 255;
      const int c16 = ch * 4 + c16i;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c16 < a.c16n && (ngrp0 + g) * NT < a.ntt) {
        const long long src = (((long long)k * a.c16n + c16) * a.ntt + (long long)(ngrp0 + g) * NT) * 64 + q;
        val = reinterpret_cast<const float4*>(a.wp)[src];
      }
      bre[v] = val;
    }
  };
  auto store_step = [&](int buf) {
    if (pre_m) {
      float* at = ldsA + (buf * TM + tm) * kATile;
#pragma unroll
      for (int j = 0; j < RJ; ++j) at[(tn * RJ + j) * kAStrW + lane] = ((pre_m >> (tn * RJ + j)) & 1u) ? pre[j] : 0.0f;
    }
    float4* bt = reinterpret_cast<float4*>(ldsB + buf * TN * kBGroup);
#pragma unroll
    for (int v = 0; v < BV; ++v) bt[tid + v * (W * 64)] = bre[v];
  };
  auto compute = [&](int buf, unsigned need, int ch) {
    if (!need) return;  // this tile has no neighbour at the step's offset
    const float* at = ldsA + (buf * TM + tm) * kATile;
    const float4* bt = reinterpret_cast<const float4*>(ldsB + (buf * TN + tn) * kBGroup);
    const int m = lane & 15, kk = lane >> 4;
    const int nc = min(4, a.c16n - ch * 4);
    auto one = [&](int i) {
      float4 b[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q) b[q] = bt[(i * 4 + q) * 64 + lane];
      const float* ap = at + m * kAStrW + i * 16 + kk;
      const float a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[q].x, acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[q].y, acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b[q].z, acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b[q].w, acc[q], 0, 0, 0);
    };
    if (nc == 4) {  // straight-line: the scheduler hoists the LDS reads of step i+1 over the MFMAs of step i
      one(0);
      one(1);
      one(2);
      one(3);
    } else {
      for (int i = 0; i < nc; ++i) one(i);
    }
  };

  // ---- pipeline over the (offset, chunk) steps of the union -----------------------------------------------------
  unsigned rem = cols;
  int c_ld = __ffs((int)rem) - 1, ch_ld = 0;      // step whose loads are issued next
  auto advance = [&]() {
    if (++ch_ld == nchunk) {
      ch_ld = 0;
      rem &= rem - 1;
      c_ld = rem ? __ffs((int)rem) - 1 : 0;
    }
  };
  load_step(c_ld, ch_ld);
  unsigned need_cur = pre_m;
  int ch_cur = ch_ld;
  store_step(0);
  advance();
  unsigned need_nxt = 0;
  int ch_nxt = 0;
  if (total_steps > 1) {
    load_step(c_ld, ch_ld);
    need_nxt = pre_m;
    ch_nxt = ch_ld;
    advance();
  }
  __syncthreads();
  for (int s = 0; s < total_steps; ++s) {
    compute(s & 1, need_cur, ch_cur);
    if (s + 1 < total_steps) {
      store_step((s + 1) & 1);   // registers loaded one step ago -> the buffer last read in step s-1
      need_cur = need_nxt;
      ch_cur = ch_nxt;
      if (s + 2 < total_steps) {
        load_step(c_ld, ch_ld);
        need_nxt = pre_m;
        ch_nxt = ch_ld;
        advance();
      }
    }
    __syncthreads();
  }

  // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int co = (n_tile0 + q) * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = __shfl(prow, (lane >> 4) * 4 + r, 64);
      if (row >= 0 && co < a.cout) a.out[(long long)row * a.cout + co] = acc[q][r];
    }
  }
}

template <int TM, int TN>
int launch_wg(const WgArgs& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)(2 * TM * kATile + 2 * TN * kBGroup) * 4 + (size_t)TM * 512 * 4 + 64;
  EFG_ALLOW_DYNAMIC_LDS((conv_wg_kernel<TM, TN>), lds);
  const unsigned gx = (unsigned)ceil_div(a.n_tiles, TM);
  const unsigned gy = (unsigned)ceil_div(a.ntt / 4, TN);
  hipLaunchKernelGGL((conv_wg_kernel<TM, TN>), dim3(gx, gy), dim3(TM * TN * 64), lds, stream, a);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" size_t efg_spconv_packed_weight_lanes_bytes(int cout, int kvol, int cin, int for_dgrad) {
  if (cout < 1 || cin < 1 || kvol < 1) return 0;
  const int red = for_dgrad ? cout : cin, nn = for_dgrad ? cin : cout;
  return (size_t)kvol * ((red + 15) / 16) * (((nn + 15) / 16 + 3) / 4 * 4) * 256 * sizeof(float);
}

extern "C" int efg_spconv_pack_weight_lanes_f32(const float* weight, int cout, int kvol, int cin, int for_dgrad,
                                                float* packed, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(cout >= 1 && cin >= 1 && kvol >= 1, "spconv: bad weight shape");
  const long long total = (long long)(efg_spconv_packed_weight_lanes_bytes(cout, kvol, cin, for_dgrad) / sizeof(float));
  hipLaunchKernelGGL(pack_weight_lanes_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0,
                     stream, weight, cout, kvol, cin, for_dgrad, packed);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

// shape: tm_tn = 10 * TM + TN with (TM, TN) in {(8,1), (4,1), (4,2), (2,2), (2,4), (1,4)}; 0 = chosen from the sizes
extern "C" int efg_spconv_forward_wg_f32(const float* in_feat, int64_t m_in, int cin, const float* packed_lanes,
                                         const float* bias, int cout, int kvol, const void* plan, int64_t m_out,
                                         int flip_offsets, int tm_tn, float* out_feat, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(cin >= 1 && cout >= 1 && kvol >= 1 && kvol <= 31, "spconv wg: bad sizes");
  if (m_out == 0) return EFG_OK;
  EFG_CHECK_ARG(m_in >= 0 && (unsigned long long)m_in * (unsigned long long)cin * 4ull < (1ull << 32),
                "spconv wg: input features exceed the 4 GB the gather addresses");
  // plan layout of spconv_tiles.hip: rows i32 [n_tiles][16] | nb i32 [n_tiles][kvol][16] | vm u32 [n_tiles][32]
  const long long n_tiles = (m_out + 1023) / 1024 * 64;
  WgArgs a;
  a.in = in_feat;
  a.wp = packed_lanes;
  a.bias = bias;
  a.rows = static_cast<const int*>(plan);
  a.nb = a.rows + n_tiles * 16;
  a.vm = reinterpret_cast<const unsigned*>(a.nb + n_tiles * kvol * 16);
  a.out = out_feat;
  a.n_tiles = n_tiles;
  a.cin = cin;
  a.cout = cout;
  a.kvol = kvol;
  a.c16n = (cin + 15) / 16;
  a.ntt = ((cout + 15) / 16 + 3) / 4 * 4;
  a.flip = flip_offsets;
  const int groups = a.ntt / 4;
  if (tm_tn == 0) tm_tn = groups >= 4 ? 24 : (groups >= 2 ? 42 : 81);
  switch (tm_tn) {
    case 81: return launch_wg<8, 1>(a, stream);
    case 41: return launch_wg<4, 1>(a, stream);
    case 42: return launch_wg<4, 2>(a, stream);
    case 22: return launch_wg<2, 2>(a, stream);
    case 24: return launch_wg<2, 4>(a, stream);
    case 14: return launch_wg<1, 4>(a, stream);
    default: set_error("spconv wg: unknown shape %d", tm_tn); return EFG_E_INVALID;
  }
}
