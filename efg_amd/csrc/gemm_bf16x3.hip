// Split-precision GEMM for the A/B arm of the bench (NOT the headline path, which stays exact fp32):
//   C[M,N] (fp32) = A[M,K] (fp32 activations) . B[K,N] (fp32 weights)  (+ bias, + ReLU)
// with every operand split into two bf16 terms, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 of the 24
// significand bits), and three bf16 MFMA products accumulated in fp32:  hi.hi + hi.lo + lo.hi  (lo.lo, 2^-16 of the
// result, is dropped).  fp32-input MFMA peaks at 157 TFLOP/s on gfx950 and bf16 at 2.5 PFLOP/s: three bf16 products cost
// 3/16 of the fp32 product, which turns the encoder's [70 688 x 256] x [256 x {256, 1024}] products from MFMA-bound
// (83 / 300 us in hipBLASLt fp32, 71-80 % of that peak) into HBM-bound ones.
// Why a kernel and not three library bf16 GEMMs: splitting the ACTIVATION in a pass of its own reads 72 MB and writes
// 72-108 MB -- as long as the fp32 product it replaces (K is only 256).  Here the activation tile is split in registers
// on its way from HBM to LDS; only the weights (256 KB) are split ahead of time, by efg_gemm_bf16x3_pack_f32, straight
// into the order in which the MFMA lanes read them.
//
// Tiling: workgroup 128 x 128 of C, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x16_bf16;
// K in steps of 32 through ONE 32 KB LDS stage (the next step's global loads are in flight in registers during the
// MFMAs): 3 workgroups per CU.  LDS holds fragment IMAGES: for every (k-step of 16, 32-row tile, hi | lo) the 64 lanes' 16
// bytes in lane order, so operand reads are conflict-free ds_read_b128 with no address arithmetic.
#include "common.h"

#include <algorithm>

namespace efg {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 128, kBK = 32;
constexpr int kStage = 16384;   // bytes of one operand's LDS image per K step: [kstep 2][tile 4][hi|lo 2][lane 64][16 B]

__device__ __forceinline__ int frag_off(int kstep, int tile, int part) { return ((kstep * 4 + tile) * 2 + part) * 1024; }

// Packed weights: [col block of 128][K step of 32] -> one 16 KB stage image (above).  The reduction does not care which k
// goes to which (k-step, lane group, element) as long as A and B agree: within a step of 32, lane group g = (k % 32) / 16
// (lanes 32 g ..), k-step (k % 16) / 8, element k % 8 -- a lane group's 16 floats of a row are contiguous.  Element (k, n)
// of B sits in lane (n % 32) + 32 g of tile (n % 128) / 32.
// (Tried on top of this layout and removed: a persistent kernel with the B slice resident in LDS (K <= 256) and A read
// straight into MFMA fragments, no LDS for A and no barrier -- 105 us against this kernel's 56 us at 70 688 x 256 x 256:
// 8 waves per CU and a 4-chunk register ring do not cover the memory latency that 12 waves and the LDS stage do.)
__global__ void __launch_bounds__(256) gemm_bf16x3_pack_kernel(const float* __restrict__ w, long long sk, long long sn, int k,
                                                               int n, int kp, int np, __bf16* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)kp * np) return;
  const int nn = (int)(i % np), kk = (int)(i / np);
  const float x = (kk < k && nn < n) ? w[kk * sk + nn * sn] : 0.0f;
  const __bf16 hi = (__bf16)x;
  const __bf16 lo = (__bf16)(x - (float)hi);
  const int cb = nn / kBN, nt = (nn % kBN) / 32, ln = nn % 32;
  const int ks = kk / kBK, kb = (kk % kBK) / 16, kstep = (kk % 16) / 8, j = kk % 8;
  const long long stage = ((long long)cb * (kp / kBK) + ks) * (kStage / 2);   // in bf16 elements
  const long long e = stage + frag_off(kstep, nt, 0) / 2 + (ln + 32 * kb) * 8 + j;
  out[e] = hi;
  out[e + 512] = lo;
}

// Both layouts an nn.Linear weight w[out, in] is needed in, in one launch: B = w^T (k = in, n = out) for y = x w^T and
// B = w (k = out, n = in) for dx = dy w.
__device__ __forceinline__ long long packed_index(int kk, int nn, int kp) {
  const int cb = nn / kBN, nt = (nn % kBN) / 32, ln = nn % 32;
  const int ks = kk / kBK, kb = (kk % kBK) / 16, kstep = (kk % 16) / 8, j = kk % 8;
  return ((long long)cb * (kp / kBK) + ks) * (kStage / 2) + frag_off(kstep, nt, 0) / 2 + (ln + 32 * kb) * 8 + j;
}

__global__ void __launch_bounds__(256) gemm_bf16x3_pack_linear_kernel(const float* __restrict__ w, int n_out, int n_in,
                                                                      __bf16* __restrict__ fwd, __bf16* __restrict__ dgrad) {
  const int ip = (n_in + 127) / 128 * 128, op = (n_out + 127) / 128 * 128;   // both dims padded to the larger granule
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)ip * op) return;
  const int ci = (int)(i % ip), ro = (int)(i / ip);
  const float x = (ro < n_out && ci < n_in) ? w[(long long)ro * n_in + ci] : 0.0f;
  const __bf16 hi = (__bf16)x;
  const __bf16 lo = (__bf16)(x - (float)hi);
  const int in_kp = (n_in + kBK - 1) / kBK * kBK, out_kp = (n_out + kBK - 1) / kBK * kBK;
  if (ci < in_kp && ro < op) {     // forward: k = ci, n = ro
    const long long e = packed_index(ci, ro, in_kp);
    fwd[e] = hi;
    fwd[e + 512] = lo;
  }
  if (ro < out_kp && ci < ip) {    // data gradient: k = ro, n = ci
    const long long e = packed_index(ro, ci, out_kp);
    dgrad[e] = hi;
    dgrad[e + 512] = lo;
  }
}

struct GemmArgs {
  const float* a;
  long long m, lda;
  int k, kp;
  const char* bp;
  int n, nb;
  const float* bias;
  int relu;
  float* c;
  long long ldc;
};

__global__ void __launch_bounds__(256) gemm_bf16x3_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kStage];   // A image, B image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long bid = blockIdx.x;
  const int cb = (int)(bid % g.nb);          // the column blocks of a row block are neighbours: its A tile is read once
  const long long row0 = (bid / g.nb) * kBM;  // from HBM and then from cache
  // A loader: per 32-row group one row per 8 threads, 16 bytes each: full 128-byte lines
  const int lr = tid >> 3, lc = tid & 7;
  const float* ap[4];
  bool rok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long row = row0 + 32 * i + lr;
    rok[i] = row < g.m;
    ap[i] = g.a + (rok[i] ? row : 0) * g.lda + lc * 4;
  }
  const int nks = g.kp / kBK;
  const char* bsrc = g.bp + (long long)cb * nks * kStage + tid * 16;
  f32x4v pa[4];
  f32x4v pb[4];   // (an ext-vector, not HIP's uint4 struct: that one ended up in scratch memory, loads waited for on issue)
  auto fetch = [&](int ks) {
    const bool kok = ks * kBK + lc * 4 < g.k;   // K is a multiple of 4 (checked by the launcher)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      pa[i] = (rok[i] && kok) ? *reinterpret_cast<const f32x4v*>(ap[i] + ks * kBK) : f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) pb[u] = *reinterpret_cast<const f32x4v*>(bsrc + (long long)ks * kStage + u * 4096);
  };
  // the thread's 4 floats of row (32 i + lr): k = 4 lc .. 4 lc + 3 of the step
  // (lane slots of the A images are permuted, slot = lane ^ 8 for lanes >= 32: the writers of lanes l and l + 32 -- the two
  // k-halves of one row -- would otherwise hit the same banks; any permutation reads conflict-free)
  const int a_kb = lc >> 2;   // floats 4 lc .. 4 lc + 3 of the step: lane group (4 lc) / 16, k-step ((4 lc) % 16) / 8
  const int a_slot = frag_off((lc >> 1) & 1, 0, 0) + ((lr + 32 * a_kb) ^ (a_kb << 3)) * 16 + (lc & 1) * 8;
  const int a_lane = (lane ^ ((lane >> 5) << 3)) * 16;
  auto stash = [&](int stage) {
    char* As = lds + stage * 2 * kStage;
    char* Bs = As + kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (__bf16)pa[i][e];
        lo[e] = (__bf16)(pa[i][e] - (float)hi[e]);
      }
      *reinterpret_cast<bf16x4*>(As + a_slot + i * 2048) = hi;
      *reinterpret_cast<bf16x4*>(As + a_slot + i * 2048 + 1024) = lo;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4v*>(Bs + u * 4096 + tid * 16) = pb[u];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // One LDS stage, two barriers per K step; step ks + 1 is in flight in registers during the MFMAs of step ks.  (Two
  // stages with one barrier per step: 64 KB, two workgroups per CU instead of three -- 56 -> 67 us; what the kernel
  // needs is waves to overlap its phases, not fewer barriers.)
  fetch(0);
  for (int ks = 0; ks < nks; ++ks) {
    stash(0);
    __syncthreads();
    if (ks + 1 < nks) fetch(ks + 1);
    const char* Ac = lds;
    const char* Bc = Ac + kStage;
#pragma unroll
    for (int kstep = 0; kstep < 2; ++kstep) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(Ac + frag_off(kstep, 2 * wm + t, 0) + a_lane);
        al[t] = *reinterpret_cast<const bf16x8*>(Ac + frag_off(kstep, 2 * wm + t, 1) + a_lane);
        bh[t] = *reinterpret_cast<const bf16x8*>(Bc + frag_off(kstep, 2 * wn + t, 0) + lane * 16);
        bl[t] = *reinterpret_cast<const bf16x8*>(Bc + frag_off(kstep, 2 * wn + t, 1) + lane * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // the two small products first, the large one last
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // C / D layout of the 32 x 32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = cb * kBN + 64 * wn + 32 * j + (lane & 31);
      if (col >= g.n) continue;
      const float b = g.bias ? g.bias[col] : 0.0f;
      const long long rbase = row0 + 64 * wm + 32 * i + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < g.m) {
          float v = acc[i][j][r] + b;
          if (g.relu) v = fmaxf(v, 0.0f);
          g.c[row * g.ldc + col] = v;
        }
      }
    }
}

// ---- weight gradient: dW[n, k] = sum_m G[m, n] . X[m, k]  (both operands row-major activations, the reduction runs
// over the ROWS) ------------------------------------------------------------------------------------------------------
// Both MFMA operands want 8 consecutive m per lane, which in a row-major matrix are a row stride apart: each loader
// thread takes an 8-row x 4-column block (8 coalesced 16-byte loads), so that after the split it holds, per column, the 8
// consecutive-m elements of one lane -- the transpose costs nothing.  Same LDS images and wave tiling as above; the M range
// is cut into chunks (one workgroup per chunk and 128 x 128 output tile) whose partial tiles a second kernel sums in
// chunk order: deterministic, no atomics.
struct WgradArgs {
  const float *g, *x;
  long long m, ldg, ldx;
  int n, k, tiles_k, rows_per_chunk;
  float* part;   // [chunks][n][k]
};

__global__ void __launch_bounds__(256) gemm_bf16x3_tn_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kStage];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x, chunk = blockIdx.y;
  const int n0 = (tile / a.tiles_k) * kBN, k0 = (tile % a.tiles_k) * kBN;
  const long long m_begin = (long long)chunk * a.rows_per_chunk;
  const long long m_end = min(m_begin + a.rows_per_chunk, a.m);
  // loader: threads 0..127 the G tile (A operand), 128..255 the X tile (B operand); 32 rows x 128 columns per stage
  const int which = tid >> 7, lt = tid & 127;
  const int mblk = lt >> 5, c4 = (lt & 31) * 4;
  const float* src = which ? a.x : a.g;
  const long long ld = which ? a.ldx : a.ldg;
  const int col0 = (which ? k0 : n0) + c4;
  const bool col_ok = col0 < (which ? a.k : a.n);   // n, k multiples of 4
  f32x4v pre[8];
  auto fetch = [&](long long mrow) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const long long row = mrow + 8 * mblk + r;
      pre[r] = (col_ok && row < m_end) ? *reinterpret_cast<const f32x4v*>(src + row * ld + col0) : f32x4v{0.f, 0.f, 0.f, 0.f};
    }
  };
  char* img = lds + which * kStage + frag_off(mblk >> 1, (lt & 31) >> 3, 0) + (((lt & 7) * 4) + 32 * (mblk & 1)) * 16;
  auto stash = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 hi, lo;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        hi[r] = (__bf16)pre[r][c];
        lo[r] = (__bf16)(pre[r][c] - (float)hi[r]);
      }
      *reinterpret_cast<bf16x8*>(img + c * 16) = hi;
      *reinterpret_cast<bf16x8*>(img + c * 16 + 1024) = lo;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const char* Ac = lds;
  const char* Bc = lds + kStage;
  if (m_begin < m_end) fetch(m_begin);
  for (long long mrow = m_begin; mrow < m_end; mrow += kBK) {
    stash();
    __syncthreads();
    if (mrow + kBK < m_end) fetch(mrow + kBK);
#pragma unroll
    for (int kstep = 0; kstep < 2; ++kstep) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(Ac + frag_off(kstep, 2 * wm + t, 0) + lane * 16);
        al[t] = *reinterpret_cast<const bf16x8*>(Ac + frag_off(kstep, 2 * wm + t, 1) + lane * 16);
        bh[t] = *reinterpret_cast<const bf16x8*>(Bc + frag_off(kstep, 2 * wn + t, 0) + lane * 16);
        bl[t] = *reinterpret_cast<const bf16x8*>(Bc + frag_off(kstep, 2 * wn + t, 1) + lane * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  float* out = a.part + (long long)chunk * a.n * a.k;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + 64 * wn + 32 * j + (lane & 31);
      if (kc >= a.k) continue;
      const int nbase = n0 + 64 * wm + 32 * i + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = nbase + (r & 3) + 8 * (r >> 2);
        if (nn < a.n) out[(long long)nn * a.k + kc] = acc[i][j][r];
      }
    }
}

__global__ void __launch_bounds__(256) gemm_bf16x3_tn_reduce_kernel(const float* __restrict__ part, int chunks, long long elems,
                                                                    float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= elems) return;
  float s = 0.0f;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {   // four loads in flight, summed in chunk order
    const float v0 = part[(long long)c * elems + i], v1 = part[(long long)(c + 1) * elems + i],
                v2 = part[(long long)(c + 2) * elems + i], v3 = part[(long long)(c + 3) * elems + i];
    s = ((s + v0) + v1) + v2 + v3;
  }
  for (; c < chunks; ++c) s += part[(long long)c * elems + i];
  out[i] = s;
}

inline int round_up(int x, int q) { return (x + q - 1) / q * q; }

// rows of a chunk: about 512 workgroups in all, whole stages of 32 rows
inline int tn_rows_per_chunk(long long m, int n, int k) {
  const int tiles = (round_up(n, kBN) / kBN) * (round_up(k, kBN) / kBN);
  const long long chunks = std::max<long long>(1, 512 / tiles);
  const long long rows = (m + chunks - 1) / chunks;
  return (int)std::max<long long>(kBK, (rows + kBK - 1) / kBK * kBK);
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" size_t efg_gemm_bf16x3_pack_bytes(int k, int n) {
  if (k < 1 || n < 1) return 0;
  return (size_t)round_up(k, kBK) * (size_t)round_up(n, kBN) * 4;   // hi + lo, 2 bytes each
}

extern "C" int efg_gemm_bf16x3_pack_f32(const float* w, int64_t stride_k, int64_t stride_n, int k, int n, void* packed,
                                        void* stream) {
  EFG_CHECK_ARG(w && packed && k >= 1 && n >= 1, "gemm_bf16x3 pack: bad arguments (k %d, n %d)", k, n);
  const int kp = round_up(k, kBK), np = round_up(n, kBN);
  const long long total = (long long)kp * np;
  hipLaunchKernelGGL(gemm_bf16x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (long long)stride_k, (long long)stride_n, k, n, kp, np, (__bf16*)packed);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_gemm_bf16x3_f32(const float* a, int64_t m, int k, int64_t lda, const void* packed_b, int n,
                                   const float* bias, int relu, float* c, int64_t ldc, void* stream) {
  EFG_CHECK_ARG(a && packed_b && c && m >= 0 && k >= 1 && n >= 1, "gemm_bf16x3: bad arguments");
  // (lda < k is allowed: rows that OVERLAP -- row r of a 3 x 3 convolution's ky-th tap block is the 3 c consecutive channels
  // starting at row r of the padded channels-last map, operators/conv2d.py; the kernel only forms a + r * lda + col)
  EFG_CHECK_ARG(k % 4 == 0 && lda % 4 == 0 && lda >= 4 && ldc >= n && ((uintptr_t)a & 15) == 0,
                "gemm_bf16x3: A rows must be 16-byte aligned with K a multiple of 4 (k %d, lda %lld)", k, (long long)lda);
  if (m == 0) return EFG_OK;
  GemmArgs g;
  g.a = a;
  g.m = m;
  g.lda = lda;
  g.k = k;
  g.kp = round_up(k, kBK);
  g.bp = (const char*)packed_b;
  g.n = n;
  g.nb = round_up(n, kBN) / kBN;
  g.bias = bias;
  g.relu = relu;
  g.c = c;
  g.ldc = ldc;
  const long long blocks = ((m + kBM - 1) / kBM) * g.nb;
  EFG_CHECK_ARG(blocks < (1ll << 31), "gemm_bf16x3: too many tiles");
  hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" size_t efg_gemm_bf16x3_wgrad_workspace_bytes(int64_t m, int n, int k) {
  if (m < 1 || n < 1 || k < 1) return 0;
  const int rows = tn_rows_per_chunk(m, n, k);
  const long long chunks = (m + rows - 1) / rows;
  return (size_t)chunks * (size_t)n * (size_t)k * sizeof(float);
}

extern "C" int efg_gemm_bf16x3_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t m, int n, int k,
                                         float* dw, void* ws, size_t ws_bytes, void* stream) {
  EFG_CHECK_ARG(g && x && dw && m >= 1 && n >= 1 && k >= 1, "gemm_bf16x3 wgrad: bad arguments");
  EFG_CHECK_ARG(n % 4 == 0 && k % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 && ldg >= n && ldx >= 4 &&   // (ldx < k: overlapping rows, as above)
                    ((uintptr_t)g & 15) == 0 && ((uintptr_t)x & 15) == 0,
                "gemm_bf16x3 wgrad: rows must be 16-byte aligned, n and k multiples of 4 (n %d, k %d)", n, k);
  const size_t need = efg_gemm_bf16x3_wgrad_workspace_bytes(m, n, k);
  EFG_CHECK_ARG(ws && ws_bytes >= need, "gemm_bf16x3 wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
  WgradArgs a;
  a.g = g;
  a.x = x;
  a.m = m;
  a.ldg = ldg;
  a.ldx = ldx;
  a.n = n;
  a.k = k;
  a.tiles_k = round_up(k, kBN) / kBN;
  a.rows_per_chunk = tn_rows_per_chunk(m, n, k);
  a.part = (float*)ws;
  const int tiles = (round_up(n, kBN) / kBN) * a.tiles_k;
  const long long chunks = (m + a.rows_per_chunk - 1) / a.rows_per_chunk;
  EFG_CHECK_ARG(chunks <= 65535, "gemm_bf16x3 wgrad: too many chunks");
  hipLaunchKernelGGL(gemm_bf16x3_tn_kernel, dim3((unsigned)tiles, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
  EFG_LAUNCH_CHECK();
  const long long elems = (long long)n * k;
  hipLaunchKernelGGL(gemm_bf16x3_tn_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)ws, (int)chunks, elems, dw);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_gemm_bf16x3_pack_linear_f32(const float* w, int n_out, int n_in, void* packed_fwd, void* packed_dgrad,
                                               void* stream) {
  EFG_CHECK_ARG(w && packed_fwd && packed_dgrad && n_out >= 1 && n_in >= 1, "gemm_bf16x3 pack_linear: bad arguments");
  const long long total = (long long)round_up(n_in, 128) * round_up(n_out, 128);
  hipLaunchKernelGGL(gemm_bf16x3_pack_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     n_out, n_in, (__bf16*)packed_fwd, (__bf16*)packed_dgrad);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
