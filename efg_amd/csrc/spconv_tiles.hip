// Sparse convolution, second generation: MASK-SORTED ROW TILES.
//
// The output-stationary kernel of spconv_conv.hip takes 16 consecutive rows (canonical spatial order) per wave and
// executes every kernel offset ANY of the 16 rows needs: measured x1.35 the useful MFMAs on the submanifold layers,
// x2.1 on the strided forwards (scripts/ubench/tile_waste.py; profiles/r02_*).  Which offsets a row needs is its
// 27-bit neighbour mask, and rows with equal masks are plentiful (ground, walls: a few hundred patterns cover a
// level) but interleaved in space.  A *tile plan* therefore re-orders the rows of a neighbour table inside chunks of
// 1024 consecutive rows by their mask (an LDS bitonic sort of {mask, row}), cuts the sorted sequence into 16-row
// tiles and stores, per tile, the rows, the neighbour columns in tile-major order and the per-offset validity masks:
// x1.10-1.15 instead of x1.35 / x2.1, with gather locality kept at chunk granularity (a chunk's gathers stay inside
// three z-slabs of ~1k rows).  The plan depends only on geometry: it is built once per rulebook on the geometry
// stream and shared by every convolution, direction and training step that uses the table (9 launches for a res18
// submanifold key).  A row's sum runs over ITS offsets only; the plan is a pure function of the table, so results are
// reproducible run to run (with KS > 1 the grouping of a row's partial sums follows its tile, as in generation one).
//
// Kernel (conv_tile_kernel): per wave R sub-tiles of 16 rows (R = 2: one B fragment load feeds two row tiles, which
// halves the weight traffic through the vector L1 -- PMC: 52 % of the L1's peak bandwidth in generation one), NT
// output-channel tiles, v_mfma_f32_16x16x4_f32, A through a wave-private LDS tile, next gather in flight during the
// MFMAs; the offsets of a tile are split over the KS waves of its workgroup and summed through LDS (small levels).
// The prologue is three coalesced loads (rows, neighbour block, masks) -- no ballots, no index arithmetic.
#include "common.h"
#include "tile_plan.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace efg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kCKt = 64;          // channels per staged chunk
constexpr int kStreamKMaxGrid = 4096;  // workgroups of a stream-K launch (256 CUs x at most 16)

// pfx[u] = items of the units before u, for units of R tiles (n_tiles is a multiple of 64).  Two workgroups, one launch:
// blockIdx.x = 0 -> R = 1 (pfx1), 1 -> R = 2 (pfx2).
__global__ void __launch_bounds__(1024) tile_prefix_kernel(const int* __restrict__ rows, const unsigned* __restrict__ vm,
                                                            long long n_tiles, int* __restrict__ pfx1, int* __restrict__ pfx2) {
  __shared__ int sm[17];
  __shared__ int carry_s;
  const int R = blockIdx.x + 1;
  int* pfx = blockIdx.x ? pfx2 : pfx1;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const long long n_units = n_tiles / R;
  for (long long base = 0; base < n_units; base += 1024) {
    const long long u = base + threadIdx.x;
    int items = 0;
    if (u < n_units) {
      unsigned cols = 0;
      bool has_rows = false;
      for (int s = 0; s < R; ++s) {
        cols |= vm[(u * R + s) * 32 + 31];
        has_rows |= rows[(u * R + s) * 16] >= 0;   // a tile's rows are sorted with the padding last
      }
      items = max(__popc(cols), has_rows ? 1 : 0);
    }
    int tot;
    const int pre = block_exclusive_scan(items, sm, &tot);
    const int carry = carry_s;
    if (u < n_units) pfx[u] = carry + pre;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) pfx[n_units] = carry_s;
}

__global__ void __launch_bounds__(256) tile_plan_kernel(const int* __restrict__ nbr, long long m, int kvol, int* __restrict__ rows,
                                                         int* __restrict__ nb, unsigned* __restrict__ vm) {
  __shared__ unsigned long long keys[kChunkRows];
  const int tid = threadIdx.x, lane = tid & 63;
  const long long c0 = (long long)blockIdx.x * kChunkRows;
#pragma unroll
  for (int q = 0; q < kChunkRows / 256; ++q) {
    const int li = tid + 256 * q;
    const long long r = c0 + li;
    unsigned long long key = ~0ull;  // padding rows sort last
    if (r < m) {
      // all (<= 31) loads of the row's table column issued before the first is used: a loop with a runtime trip count
      // waits for every load in turn (27 round trips per row; the kernel was 57 us per launch, 15 launches per step)
      int nv[31];
#pragma unroll
      for (int k = 0; k < 31; ++k) nv[k] = k < kvol ? nbr[(long long)k * m + r] : -1;
      unsigned mask = 0;
#pragma unroll
      for (int k = 0; k < 31; ++k) mask |= (nv[k] >= 0 ? 1u : 0u) << k;
      key = ((unsigned long long)mask << 10) | (unsigned)li;
    }
    keys[li] = key;
  }
  __syncthreads();
  for (int size = 2; size <= kChunkRows; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int q = 0; q < kChunkRows / 512; ++q) {
        const int t = tid + 256 * q;                                   // 512 compare-exchanges per stage
        const int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));  // lower element of pair t
        const int j = i | stride;
        const unsigned long long a = keys[i], b = keys[j];
        const bool up = (i & size) == 0;
        if ((a > b) == up) {
          keys[i] = b;
          keys[j] = a;
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int q = 0; q < kChunkRows / 256; ++q) {
    const int p = tid + 256 * q;  // sorted position inside the chunk; 16 consecutive positions = one tile
    const unsigned long long key = keys[p];
    const long long r = (key == ~0ull) ? -1 : c0 + (long long)(key & 1023u);
    const long long tile = (c0 + p) >> 4;
    const int j = p & 15;
    rows[tile * 16 + j] = (int)r;
    unsigned active = 0;
    int nv[31];
#pragma unroll
    for (int k = 0; k < 31; ++k) nv[k] = (k < kvol && r >= 0) ? nbr[(long long)k * m + r] : -1;
#pragma unroll
    for (int k = 0; k < 31; ++k) {
      if (k < kvol) {   // (wave-uniform)
        const int v = nv[k];
        nb[(tile * kvol + k) * 16 + j] = v;
        const unsigned long long bal = __ballot(v >= 0);
        const unsigned m16 = (unsigned)(bal >> (lane & 48)) & 0xffffu;
        if (j == 0) vm[tile * 32 + k] = m16;
        active |= (m16 ? 1u : 0u) << k;
      }
    }
    if (j == 0) {
      for (int k = kvol; k < 31; ++k) vm[tile * 32 + k] = 0;
      vm[tile * 32 + 31] = active;
    }
  }
}

// ---- convolution over a plan -----------------------------------------------------------------------------------
struct TileArgs {
  const float* in;       // [m_in][cin]
  const float* wp;       // packed weights (efg_spconv_pack_weight_f32 layout)
  const float* bias;     // [cout] or null
  const int* rows;       // plan
  const int* nb;
  const unsigned* vm;
  float* out;            // [m_out][cout]
  long long n_tiles;
  int cin, cout, kvol;
  int c16n, np;
  int pipe, deal, xcd;   // experiment switches (EFG_TILE_PIPE, EFG_TILE_DEAL, EFG_TILE_XCD)
  // stream-K (MODE & 2)
  const int* sk_prefix;  // [ux + 1] exclusive prefix of the units' item counts (from the plan, for this R)
  float* sk_scratch;     // [grid][R * NT * 256] shares of units that straddle a cut
  int* sk_flags;         // [grid] epoch of the last share a workgroup published
  int sk_epoch;
  int sk_polls;          // polls of a share's flag before the unit is recomputed instead (EFG_TILE_SK_POLLS, default 20000)
  int* sk_fallbacks;     // device counter: units recomputed because a share did not arrive in time (efg_spconv_streamk_fallbacks)
  unsigned ux, uy;       // the unit grid (what gridDim is otherwise)
  int bf3;               // 1: split-precision arm (MODE & 4), weights packed by efg_spconv_pack_weight_f32 with flag 4
  int flip;              // 1: offset k of the WEIGHTS reads table column kvol-1-k (dgrad of a submanifold conv:
                         // the transposed table of a symmetric window is the table with the offsets reversed)
  // PAIRS (efg_spconv_tiled_pair_f32): two convolutions over ONE table in one launch.
  //   ny1 > 0 ("N pair", forward of a stage's main + shortcut convolution): the n-slices [0, ny1) multiply by `wp` and store
  //     to `out`, the slices [ny1, 2 ny1) by `wp2` into `out2` -- per slice exactly what a launch of its own computes;
  //   in2 != null ("K pair", their data gradient): the reduction runs over the channels of `in` (weights `wp`) and then over
  //     those of `in2` (weights `wp2`), both cin wide: out = in (x) wp + in2 (x) wp2, accumulated in one pass.
  const float* in2;
  const float* wp2;
  float* out2;
  int ny1;
};

// (The 16-byte-gather variant of rounds 2-5 -- template parameter V4, EFG_TILE_V4=1: natural-order packed weights, XOR-swizzled A
// tile, one ds_read_b128 per fragment -- measured 3-13 % SLOWER on every res18 layer (profiles/r02_v4_sweep.txt) and was retired
// in round 6; the template parameter stays 0.)
#ifndef EFG_TILE_WPE_R2
#define EFG_TILE_WPE_R2 4   // waves per SIMD asked of the compiler for the stream-K R = 2 shape (A/B builds: scripts/build_ab.sh)
#endif
#ifndef EFG_TILE_WPE_R1
#define EFG_TILE_WPE_R1 5
#endif
template <int NT, int R, int KS, int V4, int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE & 2) ? (R == 2 ? EFG_TILE_WPE_R2 : EFG_TILE_WPE_R1) : 1)))
conv_tile_kernel(TileArgs a) {
  constexpr int SK = (MODE >> 1) & 1;  // stream-K: the (unit, offset) items are cut into equal shares, one per workgroup
  constexpr int WT = 4 / KS;                         // wave tiles (of R * 16 rows) per workgroup
  constexpr int BF3 = (MODE >> 2) & 1;  // A/B arm: three bf16 MFMA products of split operands instead of the fp32 MFMA
  // LDS row stride of the A tile (BF3: 16-byte aligned rows for the 32-byte fragment reads)
  constexpr int kAStr = BF3 ? kCKt + 4 : kCKt + 2;
  __shared__ __attribute__((aligned(16))) float a_tile[4][R * 16 * kAStr];  // wave-private A staging
  __shared__ int nb_tile[WT][R * 32 * 16];           // byte offsets of the neighbour rows, [sub][k][16]
  __shared__ int s_redo_flag;                        // stream-K: a share did not arrive in time, recompute the unit
  int* s_redo = &s_redo_flag;
  const int lane = threadIdx.x & 63;   // (the stream-K driver below; body() derives its own, laundered, copies)
  // j0 / j1: the ranks (among the unit's active table columns) this call multiplies, [0, 32) = all; g_first: stream-K,
  // the first workgroup that holds a share of the unit
  auto body = [&](unsigned bx, unsigned by, int j0, int j1, unsigned g_first, unsigned sk_pos) {
  // stream-K calls this in a loop: launder the lane id per call, or every lane-derived address component of the body
  // becomes a loop invariant that is hoisted and kept alive across it (+70 VGPRs: two waves per SIMD instead of four)
  int tid_l = threadIdx.x;
  if (SK) asm volatile("" : "+v"(tid_l));
  const int lane = tid_l & 63, wv = tid_l >> 6;
  const int wt = wv / KS, part = wv % KS;
  const long long t0 = ((long long)bx * WT + wt) * R;  // first 16-row tile of this wave tile
  const bool tile_ok = t0 < a.n_tiles;
  if (KS == 1 && !tile_ok) return;
  const bool second = a.ny1 > 0 && (int)by >= a.ny1;   // (N pair: this n-slice belongs to the second convolution; uniform)
  const int n_tile0 = (second ? (int)by - a.ny1 : (int)by) * NT;
  const float* wp_n = second ? a.wp2 : a.wp;
  float* out_n = second ? a.out2 : a.out;
  float* at0 = a_tile[wv];
  int* nbs = nb_tile[wt];

  // masks: lane l of vmr[s] holds vm[t0 + s][l]  (l < kvol: rows of sub-tile s with a neighbour at table column l;
  // l = 31: the sub-tile's active columns)
  unsigned vmr[R];
  int prow = -1;
  if (tile_ok) {
#pragma unroll
    for (int s = 0; s < R; ++s) vmr[s] = (lane < 32) ? a.vm[(t0 + s) * 32 + lane] : 0u;
    if (lane < R * 16) prow = a.rows[t0 * 16 + lane];
    // neighbour block of the R sub-tiles: contiguous in the plan -> coalesced; row number -> byte offset
    const int* src = a.nb + t0 * a.kvol * 16;
    const int ne = R * a.kvol * 16;
    for (int e0 = lane + 64 * part; e0 < ne; e0 += 4 * 64 * KS) {   // (four loads in flight: one round trip, not four)
      int ev[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ev[u] = src[min(e0 + u * 64 * KS, ne - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 64 * KS;
        if (e < ne) {
          const int s = e / (a.kvol * 16), rem = e - s * (a.kvol * 16);
          nbs[s * 512 + rem] = (int)((unsigned)max(ev[u], 0) * (unsigned)a.cin * 4u);
        }
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < R; ++s) vmr[s] = 0u;
  }
  unsigned cols = 0;  // table columns any sub-tile of this wave tile needs
#pragma unroll
  for (int s = 0; s < R; ++s) cols |= (unsigned)__builtin_amdgcn_readlane((int)vmr[s], 31);
  bool sk_first = true, sk_last = true;
  if (SK) {   // this workgroup's share of the unit: the active columns of rank [j0, j1)
    const int ncols = __popc(cols);
    sk_first = j0 == 0;
    sk_last = j1 >= ncols;
    unsigned sub = cols;
    for (int i = 0; i < j0; ++i) sub &= sub - 1;
    unsigned keep = 0;
    for (int i = j0; i < j1 && sub; ++i) {
      keep |= sub & (0u - sub);
      sub &= sub - 1;
    }
    cols = keep;
  }
  if (KS > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  f32x4 acc[R][NT];
#pragma unroll
  for (int s = 0; s < R; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float b = 0.0f;
      const int co = (n_tile0 + t) * 16 + (lane & 15);
      if (a.bias && co < a.cout && part == 0 && sk_first) b = a.bias[co];
      acc[s][t] = f32x4{b, b, b, b};
    }

  const int nchunk1 = (a.c16n * 16 + kCKt - 1) / kCKt;
  const int nchunk = a.in2 ? 2 * nchunk1 : nchunk1;   // (K pair: the chunks of `in`, then those of `in2`)
  float pre[R * 16];
  unsigned pre_m[R];

  auto gather = [&](int col, int ch) {
    const bool hi = __builtin_amdgcn_readfirstlane((int)(ch >= nchunk1)) != 0;   // (K pair, wave-uniform: the second operand's chunk)
    const char* src = reinterpret_cast<const char*>(hi ? a.in2 : a.in);
    const unsigned cc4 = (unsigned)min((hi ? ch - nchunk1 : ch) * kCKt + lane, a.cin - 1) * 4u;
#pragma unroll
    for (int s = 0; s < R; ++s) {
      pre_m[s] = (unsigned)__builtin_amdgcn_readlane((int)vmr[s], col);
      if (pre_m[s]) {  // wave-uniform: a sub-tile without a neighbour at this column loads nothing
        unsigned offs[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) offs[j] = (unsigned)nbs[s * 512 + col * 16 + j] + cc4;
#pragma unroll
        for (int j = 0; j < 16; ++j)
          pre[s * 16 + j] = *reinterpret_cast<const float*>(src + offs[j]);
      }
    }
  };
  auto stash = [&](float* at) {
#pragma unroll
    for (int s = 0; s < R; ++s)
      if (pre_m[s]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) at[(s * 16 + j) * kAStr + lane] = ((pre_m[s] >> j) & 1u) ? pre[s * 16 + j] : 0.0f;
      }
  };
  auto b_offset = [&](int col, int ch) {
    const int k = a.flip ? (a.kvol - 1 - col) : col;  // weight offset of this table column
    const int c16_lo = ch * (kCKt / 16);
    const int m = lane & 15, kk = lane >> 4;
    // packed weights of (k, c16, n-tile t): 16-byte fragment per lane, 1 KB per n-tile, np * 64 bytes per c16
    return ((((unsigned)k * (unsigned)a.c16n + (unsigned)c16_lo) * (unsigned)a.np + (unsigned)(n_tile0 * 16 + m)) * 16u +
            (unsigned)(kk * 4)) * 4u;
  };
  const unsigned bstep = (unsigned)a.np * 64u;
  auto load_b = [&](float4* b, const float* wbase, unsigned boff0, int i) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      b[t] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wbase) + boff0 + (unsigned)i * bstep + (unsigned)t * 1024u);
  };
  auto compute = [&](const float* at, int col, int ch, unsigned m0, unsigned m1) {
    if (BF3) {
      // Split-precision arm: per 32-channel step the lane's 8 consecutive channels of its row (two 16-byte LDS reads) are
      // split into bf16 hi / lo in registers; the weights arrive split and in lane order (2 KB per n-tile: hi | lo);
      // hi.lo + lo.hi + hi.hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate (same C layout as the fp32 form).
      const int m = lane & 15, kk = lane >> 4;
      const int k = a.flip ? (a.kvol - 1 - col) : col;
      const int c32n = a.c16n >> 1, c32_lo = ch * (kCKt / 32);
      const int nc2 = min(kCKt / 32, c32n - c32_lo);
      for (int i2 = 0; i2 < nc2; ++i2) {
        const char* bp = reinterpret_cast<const char*>(a.wp) +
                         (((size_t)k * c32n + c32_lo + i2) * (a.np >> 4) + n_tile0) * 2048u + lane * 16;
        bf16x8 bh[NT], bl[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          bh[t] = *reinterpret_cast<const bf16x8*>(bp + t * 2048);
          bl[t] = *reinterpret_cast<const bf16x8*>(bp + t * 2048 + 1024);
        }
#pragma unroll
        for (int s = 0; s < R; ++s) {
          if ((s == 0 ? m0 : m1) == 0) continue;
          const float* ap = at + (s * 16 + m) * kAStr + i2 * 32 + kk * 8;
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(ap), x1 = *reinterpret_cast<const f32x4*>(ap + 4);
          bf16x8 ah, al;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? x0[e & 3] : x1[e & 3];
            ah[e] = (__bf16)x;
            al[e] = (__bf16)(x - (float)ah[e]);
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[t], acc[s][t], 0, 0, 0);
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[t], acc[s][t], 0, 0, 0);
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[t], acc[s][t], 0, 0, 0);
          }
        }
      }
      return;
    }
    const bool hi = __builtin_amdgcn_readfirstlane((int)(ch >= nchunk1)) != 0;   // (K pair: the second operand's weights, its own chunk numbering)
    const int chl = hi ? ch - nchunk1 : ch;
    const float* wbase = hi ? a.wp2 : wp_n;
    const int c16_lo = chl * (kCKt / 16);
    const int m = lane & 15, kk = lane >> 4;
    const unsigned boff0 = b_offset(col, chl);
    const int nc = min(kCKt / 16, a.c16n - c16_lo);  // 16-channel steps of this chunk (4 unless the tail)
    auto mfmas = [&](const float4* b, int i) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if ((s == 0 ? m0 : m1) == 0) continue;  // wave-uniform: this sub-tile has no neighbour at the column
        const float* ap = at + (s * 16 + m) * kAStr + i * 16 + kk;
        const float a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t].x, acc[s][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t].y, acc[s][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b[t].z, acc[s][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b[t].w, acc[s][t], 0, 0, 0);
      }
    };
    // software pipeline over the (up to) four 16-channel steps: the weights of step i+1 are in flight during the
    // MFMAs of step i (they come from L2: ~200+ cycles, a 16-MFMA step is 512)
    float4 b0[NT], b1[NT];
    load_b(b0, wbase, boff0, 0);
    if (nc == 4 && a.pipe) {
      load_b(b1, wbase, boff0, 1);
      mfmas(b0, 0);
      load_b(b0, wbase, boff0, 2);
      mfmas(b1, 1);
      load_b(b1, wbase, boff0, 3);
      mfmas(b0, 2);
      mfmas(b1, 3);
    } else {
      for (int i = 0; i < nc; ++i) {
        if (i > 0) load_b(b0, wbase, boff0, i);
        mfmas(b0, i);
      }
    }
  };

  // The (column, channel chunk) steps of the wave tile, in order, are dealt round-robin to its KS waves: every
  // wave gets the same number of steps (+-1) whatever the number of active columns.
  // deal = 1: the (column, channel chunk) steps of the wave tile, in order, go round-robin to its KS waves (equal
  // step counts whatever the number of active columns); deal = 0: whole columns go round-robin (a wave reads the
  // consecutive chunks of the same 16 rows)
  unsigned mycols = cols;
  if (!a.deal && KS > 1) {
    mycols = 0;
    unsigned r2 = cols;
    int seen = 0;
    while (r2) {
      const int c = __ffs((int)r2) - 1;
      r2 &= r2 - 1;
      if (seen % KS == part) mycols |= 1u << c;
      ++seen;
    }
  }
  const int step_stride = a.deal ? KS : 1;
  const int total_steps = __popc(mycols) * nchunk;
  const int nsteps = a.deal ? (total_steps - part + KS - 1) / KS : total_steps;
  if (nsteps > 0) {
    unsigned rem = mycols;
    int c_cur = __ffs((int)rem) - 1, ch_cur = 0;
    auto advance = [&](int& c, int& ch, int n) {  // n steps forward in (column-major, chunk-minor) order
      ch += n;
      while (ch >= nchunk) {
        ch -= nchunk;
        rem &= rem - 1;
        c = rem ? __ffs((int)rem) - 1 : 0;
      }
    };
    if (a.deal) advance(c_cur, ch_cur, part);
    gather(c_cur, ch_cur);
    stash(at0);
    unsigned cm0 = pre_m[0], cm1 = pre_m[R - 1];
    for (int s = 0; s < nsteps; ++s) {
      int c_nxt = c_cur, ch_nxt = ch_cur;
      const bool more = (s + 1 < nsteps);
      if (more) {
        advance(c_nxt, ch_nxt, step_stride);
        gather(c_nxt, ch_nxt);
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      compute(at0, c_cur, ch_cur, cm0, cm1);
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (more) {
        stash(at0);
        cm0 = pre_m[0];
        cm1 = pre_m[R - 1];
      }
      c_cur = c_nxt;
      ch_cur = ch_nxt;
    }
  }

  if (KS > 1) {
    // partial accumulators of the other waves of this wave tile, through the (idle) A tiles: 4 n-tiles per pass
    __syncthreads();
    float* red = a_tile[wv];
#pragma unroll
    for (int s = 0; s < R; ++s)
      for (int tq = 0; tq < NT; tq += 4) {
        if (part != 0) {
#pragma unroll
          for (int t = tq; t < NT && t < tq + 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((t - tq) * 4 + r) * 64 + lane] = acc[s][t][r];
        }
        __syncthreads();
        if (part == 0) {
          for (int p = 1; p < KS; ++p) {
            const float* o = a_tile[wv + p];
#pragma unroll
            for (int t = tq; t < NT && t < tq + 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[s][t][r] += o[((t - tq) * 4 + r) * 64 + lane];
          }
        }
        __syncthreads();
      }
    if (part != 0 || !tile_ok) return;
  }
  if (SK && !(sk_first && sk_last)) {
    // A shared unit.  Every share but the last is written (raw accumulator layout, 1 KB per register row) to the
    // workgroup's slot of the scratch and published with a release store of the launch epoch; the workgroup with the
    // LAST share -- the highest index, so it only ever waits for workgroups dispatched before it -- adds the shares in
    // workgroup order (a fixed order: the result does not depend on timing) and stores the rows.
    // Coherence without fences: the shares and the flags are written and read with agent-scope RELAXED atomics (sc1
    // stores write through the XCD's L2, sc1 loads do not hit its stale lines).  A release / acquire pair would
    // write back / invalidate the whole L2 of the XCD -- per workgroup, and an acquire per poll of the flag: measured
    // 30-50x slower than the convolution itself.  Order: the share's stores have completed (s_waitcnt vmcnt(0)) before
    // the flag is stored; the reader's loads of the share are issued after it has seen the flag.
    constexpr int kSlot = R * NT * 4 * 64;
    if (!sk_last) {
      float* dst = a.sk_scratch + (size_t)sk_pos * kSlot + lane;
#pragma unroll
      for (int s = 0; s < R; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(dst + ((s * NT + t) * 4 + r) * 64, acc[s][t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(a.sk_flags + sk_pos, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    for (unsigned gp = g_first; gp < sk_pos; ++gp) {
      // A share is published by a workgroup dispatched before this one, early in its life and without waiting for
      // anybody, so this normally does not spin at all.  The bound is for a chip shared with ANOTHER process that
      // spins too (two ranks on one device): after ~20 ms the unit is recomputed here from scratch instead (s_redo;
      // the same sum in a different order), so the launch always ends.
      int ok = 1;
      if (lane == 0) {
        int polls = 0;
        while (__hip_atomic_load(a.sk_flags + gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.sk_epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (++polls > a.sk_polls) {
            ok = 0;
            break;
          }
        }
      }
      ok = __builtin_amdgcn_readfirstlane(ok);
      if (!ok) {
        if (lane == 0) {
          *s_redo = 1;
          atomicAdd(a.sk_fallbacks, 1);   // never silent: the bench line and the stream-K test read this counter
        }
        return;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      const float* src = a.sk_scratch + (size_t)gp * kSlot + lane;
#pragma unroll
      for (int s = 0; s < R; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[r] = __hip_atomic_load(src + ((s * NT + t) * 4 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[s][t][r] += v[r];
          asm volatile("" ::: "memory");   // four loads in flight at a time: the 32 of a share would cost 28 more VGPRs
        }
    }
  }
  // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int s = 0; s < R; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = (n_tile0 + t) * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = s * 16 + (lane >> 4) * 4 + r;
        const int row = __shfl(prow, j, 64);
        if (row >= 0 && co < a.cout) out_n[(long long)row * a.cout + co] = acc[s][t][r];
      }
    }
  };  // body

  if (!SK) {
    // XCD-aware order: consecutive workgroups (neighbouring sorted chunks re-read the same input rows) on one XCD
    unsigned bx = blockIdx.x, by = blockIdx.y;
    const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x, total = gridDim.x * gridDim.y, per = total >> 3;
    if (a.xcd == 1 && per > 0 && lin < (per << 3)) {          // one contiguous range of workgroups per XCD
      const unsigned nl = (lin & 7) * per + (lin >> 3);
      bx = nl % gridDim.x;
      by = nl / gridDim.x;
    } else if (a.xcd == 2 && per > 0 && lin < (per << 3)) {   // ranges of 64 workgroups dealt round-robin to the XCDs
      const unsigned blk = lin >> 9, in = lin & 511;           // 512 consecutive ids = 8 XCDs x 64
      const unsigned nl = blk * 512 + (in & 7) * 64 + (in >> 3);
      if (blk * 512 + 512 <= (per << 3)) {
        bx = nl % gridDim.x;
        by = nl / gridDim.x;
      }
    }
    body(bx, by, 0, 32, 0u, 0u);
    return;
  }
  // Stream-K.  A (16R-row tile group, n-slice) unit costs its number of active offsets (8 .. 27 on a submanifold
  // level) and the levels have 2-6 units per CU: with one unit per workgroup the launch lasts as long as the CU that
  // drew the most, heaviest units (PMC, res4: waves alive 63 % of the kernel's cycles, the matrix pipe 82 % busy
  // while they are; scripts/tile_density.py models the same 0.65-0.70).  Here the launch is as many workgroups as the
  // chip holds at once, and the list of (unit, active offset) items -- prefix sums of the units' offset counts come
  // with the plan -- is cut into equal shares.  A unit that straddles a cut is summed by the workgroup holding its last
  // share (see the epilogue of body).
  static_assert(!SK || KS == 4, "stream-K: one unit per workgroup");
  const int* P = a.sk_prefix;                       // [units_x + 1] exclusive prefix of the units' item counts
  const long long C = P[a.ux], T = C * (long long)a.uy;
  // (fewer items than workgroups: the surplus workgroups leave; every remaining one owns at least one item, so a
  // workgroup never waits for a share nobody writes)
  const unsigned G = (unsigned)min((long long)gridDim.x, T);
  if (blockIdx.x >= G) return;
  // Position of this workgroup in the item list: XCD x (= blockIdx.x & 7, the hardware deals workgroups round-robin)
  // takes the x-th contiguous eighth, so that the neighbouring chunks' input rows and a layer's weights are fetched by
  // ONE L2 instead of all eight (FETCH_SIZE per launch 58 -> 2x MB with positions = blockIdx).  The workgroup before
  // position g is then blockIdx - 8 -- dispatched earlier on the same XCD -- except at the 7 range boundaries, where
  // it is a later-dispatched one: fine while all workgroups are resident (the grid is sized for that), and covered by
  // the bounded wait otherwise.
  const unsigned per8 = G >> 3;
  const unsigned g = blockIdx.x < (per8 << 3) ? (blockIdx.x & 7) * per8 + (blockIdx.x >> 3) : blockIdx.x;
  auto lo_of = [&](unsigned gg) { return (long long)(((unsigned long long)gg * (unsigned long long)T) / G); };
  auto unit_of = [&](long long x) {                 // largest w with P[w] <= x (x < C): 64-ary search, wave-uniform
    unsigned lo = 0, hi = a.ux;                     // invariant: P[lo] <= x < P[hi]
    while (hi - lo > 1) {
      const unsigned span = hi - lo, step = (span + 63) / 64;
      const unsigned pos = min(lo + (unsigned)(lane + 1) * step, hi);
      const bool le = pos < hi ? ((long long)P[pos] <= x) : false;
      const unsigned long long bal = __ballot(le);
      const int nle = __popcll(bal);                // lanes are monotone: the first nle probes are <= x
      const unsigned nlo = nle ? min(lo + (unsigned)nle * step, hi) : lo;
      const unsigned nhi = min(lo + (unsigned)(nle + 1) * step, hi);
      lo = nlo;
      hi = nhi;
    }
    return lo;
  };
  // The share is walked from its END to its start.  The unit at the end (it continues in workgroup g + 1) comes
  // first, so its share is published early in this workgroup's life; the unit at the start -- where this workgroup may
  // hold the last share and has to collect the others -- comes last, when the workgroups before it have long
  // published theirs.  (Walked forwards, workgroup g would wait at its first unit for the LAST thing workgroup g - 1
  // does, which waits for g - 2, ...: the launch serialises -- measured 15-20x slower.)
  if (threadIdx.x == 0) *s_redo = 0;
  __syncthreads();
  const long long lo_i = lo_of(g);
  long long e = lo_of(g + 1);                       // exclusive end of what is left
  long long y = (e - 1) / C;
  unsigned w = unit_of((e - 1) - y * C);
  bool redo = false;
  while (e > lo_i) {
    const long long base = y * C + P[w];            // first item of the unit
    const int n = P[w + 1] - P[w];
    const int j1 = (int)(e - base);
    const int j0 = (int)max(0ll, lo_i - base);
    unsigned g_first = g;
    if (j0 > 0) {                                   // the workgroup that holds the unit's first item
      long long cand = (long long)(((unsigned long long)base * G) / (unsigned long long)T);
      while (cand + 1 <= (long long)g && lo_of((unsigned)cand + 1) <= base) ++cand;
      while (cand > 0 && lo_of((unsigned)cand) > base) --cand;
      g_first = (unsigned)cand;
    }
    body(w, (unsigned)y, redo ? 0 : j0, redo ? n : min(j1, n), g_first, g);
    __syncthreads();
    if (!redo && *s_redo) {                         // (uniform: read after the barrier) once more: the whole unit, alone
      __syncthreads();
      if (threadIdx.x == 0) *s_redo = 0;
      redo = true;
      continue;
    }
    redo = false;
    e = base + j0;
    if (e > lo_i) {                                 // on to the previous unit that has items (e > 0: there is one)
      do {
        if (w == 0) {
          w = a.ux;
          --y;
        }
        --w;
      } while (P[w + 1] == P[w]);
    }
    __syncthreads();  // the LDS tiles are reused by the next unit
  }
}

// ---- narrow layers (the stem): 16 or 32 (or at most 8) reduction channels, at most 32 output channels ------------------
// conv_tile_kernel gathers with one LANE per reduction channel and stages the rows in LDS: at 16 channels three lanes in four
// (at 5: eleven in twelve) load nothing, every offset costs 16 dependent loads + an LDS round trip for 4 MFMAs, and the
// launch takes ~45 us whatever the level holds (7-8 % of the HBM roof on the stem, profiles/r04q).  Here a lane owns
// (row lane & 15 of the tile, 16-byte piece lane >> 4 of each 64-byte half of the row): the 64 lanes of ONE load
// instruction fetch the 16 neighbour rows of an offset whole, a 4 x 4 transpose across the four lane groups (two
// v_permlane32_swap + two v_permlane16_swap per piece -- gfx950) turns the piece into the A operands of the offset's four
// MFMA steps (step j: channel 4 j + kk from lane group kk, the operand order of the tile kernel), and the layer's PACKED
// weights (efg_spconv_pack_weight_f32 order, <= 55 KB) sit in LDS where a lane's four B operands are one conflict-free
// 16-byte read.  No A staging, nothing shared between waves: a wave walks whole tiles with the tile's neighbour block
// and two chunks of rows in flight.
// THE SUMS ARE THE TILE KERNEL'S, BIT FOR BIT: same operand placement per MFMA, the active offsets of a tile dealt round
// robin to four accumulators (the tile kernel's four split-K waves; one accumulator below 8 offsets, as there) and added
// in the same order -- so every golden, tolerance and run-to-run digest downstream is unchanged
// (tests/test_spconv_gpu.py::test_small_kernel_bits_equal_tile_kernel).
#ifndef EFG_SMALL_KO
#define EFG_SMALL_KO 0   // knock-out builds (scripts/build_ab.sh): 1 no weight staging, 2 no row loads, 3 no neighbour loads, 4 no MFMAs, 5 no stores
#endif
__device__ __forceinline__ void transpose_pieces(float (&v)[4]) {   // lane group g, element e  <->  lane group e, element g
  // v_permlane32_swap a, b: a's lanes 32..63 <-> b's lanes 0..31; v_permlane16_swap a, b: a's odd 16-lane rows <-> b's even
  // rows (gfx950; scripts/ubench/permlane_probe.hip prints both).  Inline asm with both registers read-write: chained through
  // __builtin_amdgcn_permlane*_swap, this compiler (ROCm 7.2) returns the FIRST result for both halves of the second pair
  // (the probe's "128 of 256 wrong"); the s_nop covers the VALU-write -> permlane-read hazard the compiler would otherwise pad.
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// C16: 16-channel groups of the reduction (1 | 2); NT: 16-column output tiles (1 | 2); VEC: cin == 16 * C16 in 16-byte
// pieces, else cin <= 8 by 4-byte loads (C16 = 1; steps j = 0, 1 -- the tile kernel's steps 2 and 3 add exact zeros there).
constexpr int kSmallThreads = 256;   // (4 waves share a copy of the weights)
constexpr int kSmallKmax = 28;       // offsets of a table this kernel walks (checked by the launch code)
template <int C16, int NT, bool VEC>
__global__ void __launch_bounds__(kSmallThreads) conv_small_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // the packed weights [kvol][C16][np][kk][j] | neighbour blocks of the 4 waves
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int* nbs = reinterpret_cast<int*>(wl + (size_t)a.kvol * C16 * NT * 256) + wv * (kSmallKmax * 16);
  {
    constexpr int kPieces = (kSmallKmax * C16 * NT * 16 * 4 + kSmallThreads - 1) / kSmallThreads;   // 16-byte pieces per thread
    const int total4 = a.kvol * C16 * NT * 16 * 4;
    const float4* src = reinterpret_cast<const float4*>(a.wp);
    float4 v[kPieces];
#pragma unroll
    for (int q = 0; q < kPieces; ++q) {
      const int f4 = tid + kSmallThreads * q;
      v[q] = (f4 < total4 && EFG_SMALL_KO != 1) ? src[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < kPieces; ++q) {
      const int f4 = tid + kSmallThreads * q;
      if (f4 < total4) reinterpret_cast<float4*>(wl)[f4] = v[q];
    }
  }
  __syncthreads();
  const int i = lane & 15, kk = lane >> 4;
  constexpr int NJ = VEC ? 4 : 2;   // MFMA steps per 16-channel group
  constexpr int XS = C16 * NJ;      // A operands of an offset per lane
  const bool split = a.kvol >= 8;   // the tile kernel's KS = 4 (tile_shape)
  // XCD x (= blockIdx.x & 7) walks the x-th contiguous eighth of every pass over the tiles (neighbouring tiles gather the
  // same rows: one L2 fetches them)
  const unsigned G = gridDim.x, per8 = G >> 3;
  const unsigned g = blockIdx.x < (per8 << 3) ? (blockIdx.x & 7) * per8 + (blockIdx.x >> 3) : blockIdx.x;
  constexpr int kWaves = kSmallThreads / 64;
  const long long nw = (long long)G * kWaves;
  for (long long tile = (long long)g * kWaves + wv; tile < a.n_tiles; tile += nw) {
    const int prow = lane < 16 ? a.rows[tile * 16 + lane] : -1;
    if (__ballot(prow >= 0) == 0ull) continue;   // padding tile at the end of the plan
    const unsigned active = (unsigned)__builtin_amdgcn_readfirstlane((int)a.vm[tile * 32 + 31]);
    // the tile's neighbour block, contiguous in the plan: coalesced loads (all in flight) into the wave's LDS block
    {
      const int* src = a.nb + tile * a.kvol * 16;
      const int ne = a.kvol * 16;
      int ev[(kSmallKmax * 16 + 63) / 64];
#pragma unroll
      for (int u = 0; u < (kSmallKmax * 16 + 63) / 64; ++u)
        ev[u] = EFG_SMALL_KO == 3 ? (int)((tile * 16 + lane + u) % 1000) : src[min(lane + 64 * u, ne - 1)];
      __builtin_amdgcn_wave_barrier();   // (the previous tile's reads of the block are done)
#pragma unroll
      for (int u = 0; u < (kSmallKmax * 16 + 63) / 64; ++u)
        if (lane + 64 * u < ne) nbs[lane + 64 * u] = ev[u];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f32x4 accs[4][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = t * 16 + i;
      const float b = (a.bias && co < a.cout) ? a.bias[co] : 0.0f;
      accs[0][t] = f32x4{b, b, b, b};
#pragma unroll
      for (int q = 1; q < 4; ++q) accs[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The ACTIVE offsets of the tile in table order, four at a time: offset number r goes to accumulator r & 3 -- the tile
    // kernel's split-K wave -- so the accumulator index is static; the rows of the next four are in flight during the MFMAs
    // of these four (a wave has ~2 tiles to walk: the chain of dependent round trips per tile is its run time).
    unsigned rem = active;
    struct Group {
      int col[4];      // table column (wave-uniform), -1: none
      int id[4];       // this lane's neighbour row, -1: absent
      float x[4][XS];
    };
    auto fetch = [&](Group& gr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gr.col[q] = rem ? __ffs((int)rem) - 1 : -1;
        rem &= rem - 1;
        gr.id[q] = -1;
        if (gr.col[q] < 0) continue;   // (wave-uniform)
        gr.id[q] = nbs[gr.col[q] * 16 + i];
        const float* rp = a.in + (size_t)max(gr.id[q], 0) * a.cin;
        if (EFG_SMALL_KO == 2) {
#pragma unroll
          for (int e = 0; e < XS; ++e) gr.x[q][e] = (float)gr.id[q];
          continue;
        }
        if constexpr (VEC) {
#pragma unroll
          for (int h = 0; h < C16; ++h) {   // piece kk of the h-th 64-byte half of the row
            const float4 v4 = *reinterpret_cast<const float4*>(rp + h * 16 + kk * 4);
            gr.x[q][4 * h] = v4.x, gr.x[q][4 * h + 1] = v4.y, gr.x[q][4 * h + 2] = v4.z, gr.x[q][4 * h + 3] = v4.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) gr.x[q][j] = rp[min(4 * j + kk, a.cin - 1)];   // channel 4 j + kk (clamped; zeroed below)
        }
      }
    };
    auto step = [&](f32x4 (&acc)[NT], const float (&x)[XS], const float* wcol) {
#pragma unroll
      for (int h = 0; h < C16; ++h) {
        float4 bw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
          bw[t] = EFG_SMALL_KO == 4 ? make_float4(0.f, 0.f, 0.f, 0.f)
                                    : *reinterpret_cast<const float4*>(wcol + ((h * NT + t) * 16 + i) * 16 + kk * 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float av = x[h * NJ + j];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float bv = j == 0 ? bw[t].x : j == 1 ? bw[t].y : j == 2 ? bw[t].z : bw[t].w;
            if (EFG_SMALL_KO == 4) acc[t][0] += av;
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    };
    auto multiply = [&](Group& gr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (gr.col[q] < 0) continue;   // (wave-uniform)
        float xv[XS];
#pragma unroll
        for (int e = 0; e < XS; ++e) xv[e] = gr.x[q][e];
        if constexpr (VEC) {
#pragma unroll
          for (int h = 0; h < C16; ++h) {
            float pc[4] = {xv[4 * h], xv[4 * h + 1], xv[4 * h + 2], xv[4 * h + 3]};
            transpose_pieces(pc);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[4 * h + j] = pc[j];
          }
        }
#pragma unroll
        for (int e = 0; e < XS; ++e)
          if (gr.id[q] < 0 || (!VEC && 4 * (e % NJ) + kk >= a.cin)) xv[e] = 0.0f;
        const int k = a.flip ? a.kvol - 1 - gr.col[q] : gr.col[q];
        const float* wcol = wl + (size_t)k * C16 * NT * 256;
        if (split || q == 0) step(accs[q], xv, wcol);
        else step(accs[0], xv, wcol);   // fewer than 8 offsets: the tile kernel's waves do not split them
      }
    };
    Group ga, gb;
    fetch(ga);
    while (true) {
      fetch(gb);
      multiply(ga);
      if (gb.col[0] < 0) break;
      fetch(ga);
      multiply(gb);
      if (ga.col[0] < 0) break;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {   // the tile kernel's order: the first wave adds the second's, the third's, the fourth's
      acc[t] = accs[0][t];
#pragma unroll
      for (int q = 1; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += accs[q][t][r];
    }
    // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = __shfl(prow, kk * 4 + r, 64);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int co = t * 16 + i;
        if (row >= 0 && co < a.cout && (EFG_SMALL_KO != 5 || acc[t][r] == 123.456f)) a.out[(long long)row * a.cout + co] = acc[t][r];
      }
    }
  }
}

// workgroups of `kernel` one CU holds at once (LDS / VGPR bound), x the CUs of the device
template <typename K>
int resident_workgroups(K kernel) {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    cus = 256;
  return per_cu * cus;
}

template <int NT, int R, int KS, int V4, int MODE>
void launch_tiles_k(TileArgs a, unsigned gx, unsigned gy, hipStream_t stream) {
  if (MODE & 2) {
    static const int resident = resident_workgroups(conv_tile_kernel<NT, R, KS, V4, MODE>);
    a.ux = gx;
    a.uy = gy;
    hipLaunchKernelGGL((conv_tile_kernel<NT, R, KS, V4, MODE>), dim3((unsigned)std::min(resident, kStreamKMaxGrid)), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((conv_tile_kernel<NT, R, KS, V4, MODE>), dim3(gx, gy), dim3(256), 0, stream, a);
  }
}

template <int NT, int R, int V4, int MODE>
void launch_tiles_x(const TileArgs& a, int ny, int ks, hipStream_t stream) {
  const long long wave_tiles = (a.n_tiles + R - 1) / R;
  // (stream-K exists for the split-K shape only: one unit per workgroup)
  if (ks == 4) launch_tiles_k<NT, R, 4, V4, MODE>(a, (unsigned)wave_tiles, ny, stream);
  else if (ks == 2) launch_tiles_k<NT, R, 2, V4, 0>(a, (unsigned)ceil_div(wave_tiles, 2), ny, stream);
  else launch_tiles_k<NT, R, 1, V4, 0>(a, (unsigned)ceil_div(wave_tiles, 4), ny, stream);
}

template <int NT, int R, int V4>
void launch_tiles_v(const TileArgs& a, int ny, int ks, hipStream_t stream) {
  // stream-K exists for the 64-channel-wide wave tiles with 4-byte gathers and the split-K shape only (the default shapes)
  constexpr bool kModes = (NT == 4 && V4 == 0);
  if (kModes && a.bf3 && ks == 4) {   // the split-precision arm exists for this shape only (run_tiles checked)
    if (a.sk_scratch) launch_tiles_k<NT, R, 4, V4, kModes ? 6 : 0>(a, (unsigned)((a.n_tiles + R - 1) / R), ny, stream);
    else launch_tiles_k<NT, R, 4, V4, kModes ? 4 : 0>(a, (unsigned)((a.n_tiles + R - 1) / R), ny, stream);
    return;
  }
  if (kModes && a.sk_scratch && ks == 4) launch_tiles_x<NT, R, V4, kModes ? 2 : 0>(a, ny, ks, stream);
  else launch_tiles_x<NT, R, V4, 0>(a, ny, ks, stream);
}

template <int NT, int R>
void launch_tiles(const TileArgs& a, int ny, int ks, hipStream_t stream) {
  launch_tiles_v<NT, R, 0>(a, ny, ks, stream);
}

// The launch shape of a (cin, cout, kvol, rows) convolution: NT n-tiles per wave, R sub-tiles per wave, KS split-K
// waves, pipelined weight loads.  ONE definition for the launcher below and for efg_spconv_tile_shape (what the host
// side labels its timings with).
// EFG_TILE_STREAMK: 0 off, 1 (default) where it wins, 2 every eligible shape (tests)
int streamk_mode() {
  static const int v = getenv("EFG_TILE_STREAMK") ? atoi(getenv("EFG_TILE_STREAMK")) : 1;
  return v;
}

void tile_shape(int cin, int cout, int kvol, int64_t m_in, int64_t m_out, int* nt_out, int* r_out, int* ks_out, int* pipe_out) {
  constexpr int r_env = 0;
  constexpr int ks_env = 0;
  constexpr int nt_env = 0;
  constexpr int pipe_env = -1;
  const int ntiles = (cout + 15) / 16;
  int nt = ntiles >= 4 ? 4 : (ntiles >= 2 ? 2 : 1);
  if (nt_env > 0) nt = nt_env <= 1 ? 1 : (nt_env <= 2 ? 2 : 4);
  // two row sub-tiles per wave (weights loaded once for 32 rows) on the dense tables of the submanifold layers: from
  // 128 reduction channels, and from 64 when the launch is stream-K (the longer units no longer unbalance the launch:
  // 64-channel layer 105 -> 99 us forward, 104 -> 96 us dgrad; without stream-K R = 2 loses there, 107 -> 121 us)
  int r = (nt >= 4 && kvol >= 8 && m_in == m_out && (cin >= 128 || (streamk_mode() && cin >= 64))) ? 2 : 1;
  if (r_env > 0) r = r_env >= 2 ? 2 : 1;
  int ks = (kvol >= 8) ? 4 : 1;
  if (ks_env > 0) ks = ks_env >= 4 ? 4 : (ks_env >= 2 ? 2 : 1);
  *nt_out = nt;
  *r_out = r;
  *ks_out = ks;
  *pipe_out = pipe_env >= 0 ? pipe_env : (r == 2 ? 1 : 0);
}

// Stream-K scratch per (device, stream): kernels of one stream are serialised, so they can share it.  Allocated on first
// use, kept for the life of the process (a few MB).  The flags are monotonic: a launch publishes its own epoch.
struct StreamKState {
  float* scratch = nullptr;
  int* flags = nullptr;      // [kStreamKMaxGrid] epochs + [1] fallback counter
  int epoch = 0;
};

std::mutex g_sk_mu;
std::map<std::pair<int, hipStream_t>, StreamKState> g_sk_pool;

int streamk_for_stream(hipStream_t stream, StreamKState** out) {
  auto& pool = g_sk_pool;
  int dev = 0;
  EFG_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_sk_mu);
  auto key = std::make_pair(dev, stream);
  auto it = pool.find(key);
  if (it == pool.end()) {
    StreamKState st;
    EFG_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st.scratch), (size_t)kStreamKMaxGrid * 2 * 4 * 256 * sizeof(float)));
    EFG_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st.flags), (size_t)(kStreamKMaxGrid + 1) * sizeof(int)));
    EFG_HIP_TRY(hipMemset(st.flags, 0, (size_t)(kStreamKMaxGrid + 1) * sizeof(int)));
    it = pool.emplace(key, st).first;
  }
  *out = &it->second;
  return EFG_OK;
}

// The split-precision arm covers the 64-channel-wide split-K shape (NT = 4, KS = 4) with whole 32-channel reduction steps
// and whole groups of 4 n-tiles; ONE rule for the launcher, the packer's caller and efg_spconv_tile_bf16x3_ok.
bool bf16x3_ok(int cin, int cout, int kvol, int64_t m_in, int64_t m_out) {
  int nt, r, ks, pipe;
  tile_shape(cin, cout, kvol, m_in, m_out, &nt, &r, &ks, &pipe);
  return nt == 4 && ks == 4 && kvol <= 31 && cin >= 64 && cin % 32 == 0 && cout % 64 == 0;
}

// conv_small_kernel covers a (cin -> cout, kvol) layer with 16 or 32 reduction channels (16-byte pieces) or at most 8 (4-byte
// loads), at most 32 outputs and 28 offsets, whose packed weights fit 56 KB of LDS.  ONE rule for the launcher and for
// efg_spconv_small_ok (the host's labels).  c16: 16-channel groups of the reduction; nt: 16-column output tiles.
// EFG_CONV_SMALL=0 (read per call: tests flip it in-process) keeps every layer on conv_tile_kernel.
bool small_shape(int cin, int cout, int kvol, int* c16_out, int* nt_out) {
  const char* env = getenv("EFG_CONV_SMALL");
  const int c16 = (cin + 15) / 16, nt = (cout + 15) / 16;
  *c16_out = c16;
  *nt_out = nt;
  // (32 -> 32 would need 168 VGPRs + spills: it stays on the tile kernel)
  return !(env && atoi(env) == 0) && (cin == 16 || cin == 32 || (cin >= 1 && cin <= 8)) && nt <= 2 && c16 * nt <= 2 && kvol >= 1 &&
         kvol <= kSmallKmax && (size_t)kvol * c16 * nt * 1024 <= 56 * 1024;
}

int run_tiles(const float* in, int64_t m_in, int cin, const float* wp, const float* bias, int cout, int kvol,
              const void* plan, int64_t m_out, float* out, int flip, int natural_order, int bf3, hipStream_t stream,
              const float* in2 = nullptr, const float* wp2 = nullptr, float* out2 = nullptr) {
  EFG_CHECK_ARG(cin >= 1 && cout >= 1, "spconv tiled: bad channel counts");
  // pairs (TileArgs): wp2 + out2 = two convolutions of one input, wp2 + in2 = one convolution of two inputs
  EFG_CHECK_ARG(!wp2 || ((in2 != nullptr) != (out2 != nullptr) && !bias && !natural_order && !bf3),
                "spconv tiled pair: second weights need either a second input or a second output, no bias, the fp32 4-byte path");
  EFG_CHECK_ARG(wp2 || (!in2 && !out2), "spconv tiled pair: second input / output without second weights");
  EFG_CHECK_ARG(!bf3 || (!natural_order && bf16x3_ok(cin, cout, kvol, m_in, m_out)),
                "spconv tiled: the bf16x3 arm does not cover %d -> %d channels, kvol %d (ask efg_spconv_tile_bf16x3_ok)", cin,
                cout, kvol);
  EFG_CHECK_ARG(kvol >= 1 && kvol <= 31, "spconv tiled: kernel volume must be in [1,31], got %d", kvol);
  if (m_out == 0) return EFG_OK;
  EFG_CHECK_ARG(m_in >= 0 && (unsigned long long)m_in * (unsigned long long)cin * 4ull < (1ull << 32),
                "spconv tiled: input features of %lld x %d floats exceed the 4 GB the gather addresses", (long long)m_in, cin);
  const PlanView pv = plan_view(const_cast<void*>(plan), m_out, kvol);
  TileArgs a;
  a.in = in;
  a.wp = wp;
  a.bias = bias;
  a.rows = pv.rows;
  a.nb = pv.nb;
  a.vm = pv.vm;
  a.out = out;
  a.n_tiles = pv.n_tiles;
  a.cin = cin;
  a.cout = cout;
  a.kvol = kvol;
  a.c16n = (cin + 15) / 16;
  a.np = (cout + 15) / 16 * 16;
  a.flip = flip;
  a.bf3 = bf3 ? 1 : 0;
  a.in2 = in2;
  a.wp2 = wp2;
  a.out2 = out2;
  a.ny1 = 0;
  EFG_CHECK_ARG(!natural_order, "spconv tiled: the natural-order (16-byte gather) variant was retired in round 6 (3-13 %% slower, profiles/r02_v4_sweep.txt)");
  // narrow layers (conv_small_kernel): EFG_CONV_SMALL=0 keeps them on the tile kernel (A/B)
  {
    int c16 = 0, nt_s = 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(wp) & 15) == 0;
    if (small_shape(cin, cout, kvol, &c16, &nt_s) && !a.bf3 && aligned && !wp2) {
      const size_t lds = (size_t)kvol * c16 * nt_s * 1024 + (size_t)(kSmallThreads / 64) * kSmallKmax * 16 * sizeof(int);
      // as many workgroups as the device holds at once (up to 4 per CU; 2 with 55 KB of weights), a multiple of 8
      const long long fit = std::max<long long>(1, std::min<long long>(4, (150 * 1024) / (long long)lds)) * 256;
      const unsigned wgs = (unsigned)std::min<long long>(fit, std::max<long long>(8, ceil_div(pv.n_tiles, 4) / 8 * 8));
#define EFG_SMALL_LAUNCH(C, NTT, VE) \
  hipLaunchKernelGGL((conv_small_kernel<C, NTT, VE>), dim3(wgs), dim3(kSmallThreads), lds, stream, a)
      if (cin <= 8) {
        if (nt_s == 1) EFG_SMALL_LAUNCH(1, 1, false);
        else EFG_SMALL_LAUNCH(1, 2, false);
      } else if (c16 == 1) {
        if (nt_s == 1) EFG_SMALL_LAUNCH(1, 1, true);
        else EFG_SMALL_LAUNCH(1, 2, true);
      } else {
        EFG_SMALL_LAUNCH(2, 1, true);
      }
#undef EFG_SMALL_LAUNCH
      EFG_LAUNCH_CHECK();
      return EFG_OK;
    }
  }
  const int ntiles = a.np / 16;
  // R = 2 sub-tiles per wave (weights loaded once for 32 rows) once the level has enough row tiles to fill the chip
  // that way; n-tiles per wave as many as the grid allows (A reuse); offsets split over the 4 waves of a workgroup
  // (KS) on the small levels, where a wave would otherwise walk ~20 offsets x 4 channel chunks alone.
  // Launch shape (scripts/tile_sweep.sh, profiles/r02_tile_sweep.txt).  These kernels are latency-bound rather than
  // MFMA-bound (PMC: matrix pipe 50-60 % busy with 2-4 waves per SIMD), so the shape maximises waves in flight:
  //  * KS = 4: the (offset, chunk) steps of a row tile are dealt to the 4 waves of its workgroup (3x3x3 windows);
  //  * R = 2 sub-tiles per wave + software-pipelined weight loads once a step is long enough to amortise the extra
  //    registers (submanifold layers with cin >= 128; the strided layers' sparse tables lose with it), else R = 1;
  //  * NT = min(4, n-tiles): 64 output channels per wave, wider outputs tile over grid.y (each re-gathers A).
  int nt, r, ks, pipe;
  tile_shape(cin, cout, kvol, m_in, m_out, &nt, &r, &ks, &pipe);
  int ny = (ntiles + nt - 1) / nt;
  if (out2) {   // N pair: the second convolution's n-slices follow the first's in the unit grid
    a.ny1 = ny;
    ny *= 2;
  }
  constexpr int deal_env = 1;
  a.pipe = pipe;
  a.deal = deal_env;
  constexpr int xcd_env = 1;
  a.xcd = xcd_env;
  // Stream-K (see the kernel): on by default where it was measured to win -- the 64-channel-wide split-K shape on
  // submanifold tables (every level: -3 % at 64 channels, -5 % at 128, -9 % at 256) and on the strided tables with
  // at least 128 channels on both sides (-11 %); the other strided tables lose with it (short units, many of them).
  const int sk_env = streamk_mode();
  static const int sk_polls_env = getenv("EFG_TILE_SK_POLLS") ? atoi(getenv("EFG_TILE_SK_POLLS")) : 20000;
  a.sk_prefix = nullptr;
  a.sk_scratch = nullptr;
  a.sk_flags = nullptr;
  a.sk_epoch = 0;
  a.ux = a.uy = 0;
  a.sk_polls = sk_polls_env;
  a.sk_fallbacks = nullptr;
  // A launch that is being CAPTURED into a HIP graph must not use stream-K: the epoch is a kernel argument, every replay
  // would carry the same value, find the flags already equal to it and sum stale shares without waiting; the first-use
  // hipMalloc / hipMemset would invalidate the capture as well.  Captured launches take the plain (non-split) path.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (stream) (void)hipStreamIsCapturing(stream, &cap);
  if (cap == hipStreamCaptureStatusNone && sk_env && nt == 4 && ks == 4 && (sk_env == 2 || m_in == m_out || (cin >= 128 && cout >= 128))) {
    StreamKState* st = nullptr;
    if (int rc = streamk_for_stream(stream, &st)) return rc;
    a.sk_prefix = r == 2 ? pv.pfx2 : pv.pfx1;
    a.sk_scratch = st->scratch;
    a.sk_flags = st->flags;
    a.sk_epoch = ++st->epoch;   // (under the caller's serialisation of the stream)
    a.sk_fallbacks = st->flags + kStreamKMaxGrid;
  }
  if (r == 2) {
    switch (nt) {
      case 4: launch_tiles<4, 2>(a, ny, ks, stream); break;
      case 2: launch_tiles<2, 2>(a, ny, ks, stream); break;
      default: launch_tiles<1, 2>(a, ny, ks, stream); break;
    }
  } else {
    switch (nt) {
      case 4: launch_tiles<4, 1>(a, ny, ks, stream); break;
      case 2: launch_tiles<2, 1>(a, ny, ks, stream); break;
      default: launch_tiles<1, 1>(a, ny, ks, stream); break;
    }
  }
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" size_t efg_spconv_tile_plan_bytes(int64_t m, int kvol) {
  if (m < 0 || kvol < 1 || kvol > 31) return 0;
  return plan_bytes(m, kvol) + 256;
}

extern "C" int efg_spconv_tile_plan(const int32_t* nbr, int64_t m, int kvol, void* plan, size_t plan_bytes_given,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(m >= 0 && m < (1ll << 31) && kvol >= 1 && kvol <= 31, "tile_plan: bad sizes (m=%lld, kvol=%d)", (long long)m, kvol);
  if (m == 0) return EFG_OK;
  EFG_CHECK_ARG(nbr && plan && plan_bytes_given >= plan_bytes(m, kvol), "tile_plan: null pointer / plan buffer too small");
  const PlanView pv = plan_view(plan, m, kvol);
  hipLaunchKernelGGL(tile_plan_kernel, dim3((unsigned)ceil_div(m, kChunkRows)), dim3(256), 0, stream, nbr, (long long)m, kvol,
                     pv.rows, pv.nb, pv.vm);
  hipLaunchKernelGGL(tile_prefix_kernel, dim3(2), dim3(1024), 0, stream, pv.rows, pv.vm, pv.n_tiles, pv.pfx1, pv.pfx2);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_spconv_tile_shape(int cin, int cout, int kvol, int64_t m_in, int64_t m_out, int* nt, int* r, int* ks) {
  EFG_CHECK_ARG(cin >= 1 && cout >= 1 && kvol >= 1 && nt && r && ks, "tile_shape: bad arguments");
  int pipe = 0;
  tile_shape(cin, cout, kvol, m_in, m_out, nt, r, ks, &pipe);
  return EFG_OK;
}

extern "C" int efg_spconv_forward_tiled_f32(const float* in_feat, int64_t m_in, int cin, const float* packed_weight,
                                            const float* bias, int cout, int kvol, const void* plan, int64_t m_out,
                                            int flip_offsets, float* out_feat, void* stream) {
  return run_tiles(in_feat, m_in, cin, packed_weight, bias, cout, kvol, plan, m_out, out_feat, flip_offsets & 1, (flip_offsets >> 1) & 1,
                   (flip_offsets >> 2) & 1, (hipStream_t)stream);
}

extern "C" int efg_spconv_tiled_pair_f32(const float* in_a, const float* in_b, int64_t m_in, int cin, const float* packed_a,
                                         const float* packed_b, int cout, int kvol, const void* plan, int64_t m_out,
                                         int flip_offsets, float* out_a, float* out_b, void* stream) {
  EFG_CHECK_ARG(packed_a && packed_b && in_a && out_a, "spconv tiled pair: null pointer");
  EFG_CHECK_ARG((flip_offsets & ~1) == 0, "spconv tiled pair: only the offset-flip flag is accepted");
  return run_tiles(in_a, m_in, cin, packed_a, nullptr, cout, kvol, plan, m_out, out_a, flip_offsets & 1, 0, 0, (hipStream_t)stream,
                   in_b, packed_b, out_b);
}

extern "C" int efg_spconv_streamk_fallbacks(int64_t* count_out, int reset) {
  EFG_CHECK_ARG(count_out != nullptr, "streamk_fallbacks: null output");
  int dev = 0;
  EFG_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_sk_mu);
  int64_t total = 0;
  for (auto& kv : g_sk_pool) {
    if (kv.first.first != dev) continue;
    int v = 0;
    EFG_HIP_TRY(hipMemcpy(&v, kv.second.flags + kStreamKMaxGrid, sizeof(int), hipMemcpyDeviceToHost));  // (synchronises)
    total += v;
    if (reset && v) EFG_HIP_TRY(hipMemset(kv.second.flags + kStreamKMaxGrid, 0, sizeof(int)));
  }
  *count_out = total;
  return EFG_OK;
}

extern "C" int efg_spconv_small_ok(int cin, int cout, int kvol, int* c16, int* nt) {
  int s_ = 0, n_ = 0;
  const bool ok = cin >= 1 && cout >= 1 && kvol >= 1 && small_shape(cin, cout, kvol, &s_, &n_);
  if (c16) *c16 = s_;
  if (nt) *nt = n_;
  return ok ? 1 : 0;
}

extern "C" int efg_spconv_tile_bf16x3_ok(int cin, int cout, int kvol, int64_t m_in, int64_t m_out) {
  return (cin >= 1 && cout >= 1 && kvol >= 1 && bf16x3_ok(cin, cout, kvol, m_in, m_out)) ? 1 : 0;
}
