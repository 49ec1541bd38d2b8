// Self-attention over short sequences, exact fp32 on the matrix cores: softmax(Q K^T / sqrt(d)) V and its backward for
// sequences of at most 128 tokens with 64-wide heads.
//
// TrajectoryFormer's point encoder (reference: $TF/modules/transformer.py:44-92, nn.MultiheadAttention over the 128
// points of every trajectory hypothesis) is 1232 x 4 independent 128 x 128 attentions per layer and step.  One
// (sequence, head) fits on a CU: a workgroup of four waves holds K and V (2 x 34 KB of LDS), every wave owns 32
// queries.  The wave computes S^T = K Q^T, so that a 16 x 16 accumulator tile holds, per lane, four consecutive KEYS of
// one QUERY -- which is exactly the A-operand layout of v_mfma_f32_16x16x4_f32 for the next product (P V in the
// forward, dS K in the backward): the probabilities never leave the registers, there is no transpose through LDS, and
// the row reductions of the softmax are 32 in-lane values + two cross-lane steps.  The backward recomputes the
// probabilities from the saved log-sum-exp in two kernels: dQ with waves owning queries (S^T orientation), dK / dV
// with waves owning keys (S orientation, where the tile is the A operand of P^T dO and dS^T Q).  Q, K, V are read
// straight from the fused in-projection output [B, S, 3, H, 64]; dQ, dK, dV are written into ONE tensor of the same
// layout, the gradient of that projection.
//
// k-index permutation: the dot products over d use lane group j for d = 16 t + 4 j + s (s = MFMA step), so that one
// 16-byte LDS / global read feeds four MFMAs; A and B use the same permutation, the sum is order-independent.
#include "common.h"

namespace efg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kS = 128;   // tokens per sequence, at most
constexpr int kD = 64;    // head width
constexpr int kLd = 68;   // LDS row stride (floats): 16-byte aligned rows, 4-bank skew between rows
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ f32x4 mfma_k4(const f32x4 a, const f32x4 b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

// rows [0, seq) of a [seq, 64] strided matrix into LDS, rows beyond zero-filled
__device__ __forceinline__ void load_rows(float (*dst)[kLd], const float* __restrict__ src, int seq, long long stride) {
  for (int idx = threadIdx.x; idx < kS * 16; idx += 256) {
    const int row = idx >> 4, c4 = idx & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < seq) v = *reinterpret_cast<const f32x4*>(src + row * stride + 4 * c4);
    *reinterpret_cast<f32x4*>(&dst[row][4 * c4]) = v;
  }
}

// B-operand fragments of 2 x 16 rows starting at row0: frag[tile][t] = src[row0 + 16 tile + (lane & 15)][16 t + 4 j ..]
__device__ __forceinline__ void load_frags(f32x4 (&frag)[2][4], const float* __restrict__ src, int row0, int seq, long long stride,
                                           float mul) {
  const int lane = threadIdx.x & 63, j = lane >> 4, c = lane & 15;
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    const int row = row0 + 16 * tile + c;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < seq) v = *reinterpret_cast<const f32x4*>(src + row * stride + 16 * t + 4 * j);
      frag[tile][t] = v * mul;
    }
  }
}

__device__ __forceinline__ float group_max(float v) {   // over the four lane groups j that share a column
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// ---- forward ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(const float* __restrict__ qkv, int seq, int heads, float scale,
                                                          float* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) float Ks[kS][kLd];
  __shared__ __attribute__((aligned(16))) float Vs[kS][kLd];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long long rs = 3LL * heads * kD;
  const float* base = qkv + (long long)b * seq * rs + h * kD;
  load_rows(Ks, base + heads * kD, seq, rs);
  load_rows(Vs, base + 2 * heads * kD, seq, rs);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int q0 = 32 * wv;
  f32x4 bq[2][4];
  load_frags(bq, base, q0, seq, rs, scale * kLog2e);
  __syncthreads();
  if (q0 >= seq) return;

  // S^T tiles: acc[kt][qt][i] = log2e * scale * <K[16 kt + 4 j + i], Q[q0 + 16 qt + c]>
  f32x4 acc[8][2];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    acc[kt][0] = acc[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (16 * kt < seq) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&Ks[16 * kt + c][16 * t + 4 * j]);
        acc[kt][0] = mfma_k4(a, bq[0][t], acc[kt][0]);
        acc[kt][1] = mfma_k4(a, bq[1][t], acc[kt][1]);
      }
    }
  }
  // softmax over the keys of each query column
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = (16 * kt + 4 * j + i < seq) ? acc[kt][qt][i] : -INFINITY;
        acc[kt][qt][i] = v;
        m = fmaxf(m, v);
      }
    m = group_max(m);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = exp2f(acc[kt][qt][i] - m);
        acc[kt][qt][i] = p;
        sum += p;
      }
    sum = group_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) acc[kt][qt] *= inv;
    const int q = q0 + 16 * qt + c;
    if (j == 0 && q < seq) lse[((long long)b * heads + h) * seq + q] = (m + log2f(sum)) * kLn2;
  }
  // O = P V: the S^T tile is the A operand (row = query c, k = key 4 j + s)
  f32x4 o[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    if (16 * kt < seq) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 v;
        v.x = Vs[16 * kt + 4 * j + 0][16 * dt + c];
        v.y = Vs[16 * kt + 4 * j + 1][16 * dt + c];
        v.z = Vs[16 * kt + 4 * j + 2][16 * dt + c];
        v.w = Vs[16 * kt + 4 * j + 3][16 * dt + c];
        o[0][dt] = mfma_k4(acc[kt][0], v, o[0][dt]);
        o[1][dt] = mfma_k4(acc[kt][1], v, o[1][dt]);
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + 16 * qt + 4 * j + i;
      if (q < seq) {
        float* dst = out + (((long long)b * seq + q) * heads + h) * kD + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dst[16 * dt] = o[qt][dt][i];
      }
    }
}

// ---- backward: dQ (waves own queries) ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                             const float* __restrict__ lse, const float* __restrict__ dout,
                                                             int seq, int heads, float scale, float* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) float Ks[kS][kLd];
  __shared__ __attribute__((aligned(16))) float Vs[kS][kLd];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long long rs = 3LL * heads * kD, os = (long long)heads * kD;
  const float* base = qkv + (long long)b * seq * rs + h * kD;
  load_rows(Ks, base + heads * kD, seq, rs);
  load_rows(Vs, base + 2 * heads * kD, seq, rs);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int q0 = 32 * wv;
  f32x4 bq[2][4], bdo[2][4];
  load_frags(bq, base, q0, seq, rs, scale * kLog2e);
  load_frags(bdo, dout + (long long)b * seq * os + h * kD, q0, seq, os, 1.f);
  float delta[2], l2[2];
  {
    f32x4 bo[2][4];
    load_frags(bo, out + (long long)b * seq * os + h * kD, q0, seq, os, 1.f);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 pr = bo[qt][t] * bdo[qt][t];
        part += pr.x + pr.y + pr.z + pr.w;
      }
      delta[qt] = group_sum(part);
      const int q = q0 + 16 * qt + c;
      l2[qt] = q < seq ? lse[((long long)b * heads + h) * seq + q] * kLog2e : 0.f;
    }
  }
  __syncthreads();
  if (q0 >= seq) return;

  f32x4 dq[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int kt = 0; kt < 8; ++kt) {
    if (16 * kt >= seq) break;
    f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 ak = *reinterpret_cast<const f32x4*>(&Ks[16 * kt + c][16 * t + 4 * j]);
      const f32x4 av = *reinterpret_cast<const f32x4*>(&Vs[16 * kt + c][16 * t + 4 * j]);
      s[0] = mfma_k4(ak, bq[0][t], s[0]);
      s[1] = mfma_k4(ak, bq[1][t], s[1]);
      dp[0] = mfma_k4(av, bdo[0][t], dp[0]);
      dp[1] = mfma_k4(av, bdo[1][t], dp[1]);
    }
    f32x4 ds[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = (16 * kt + 4 * j + i < seq) ? exp2f(s[qt][i] - l2[qt]) : 0.f;
        ds[qt][i] = p * (dp[qt][i] - delta[qt]);
      }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 kv;
      kv.x = Ks[16 * kt + 4 * j + 0][16 * dt + c];
      kv.y = Ks[16 * kt + 4 * j + 1][16 * dt + c];
      kv.z = Ks[16 * kt + 4 * j + 2][16 * dt + c];
      kv.w = Ks[16 * kt + 4 * j + 3][16 * dt + c];
      dq[0][dt] = mfma_k4(ds[0], kv, dq[0][dt]);
      dq[1][dt] = mfma_k4(ds[1], kv, dq[1][dt]);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + 16 * qt + 4 * j + i;
      if (q < seq) {
        float* dst = dqkv + ((long long)b * seq + q) * rs + h * kD + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dst[16 * dt] = dq[qt][dt][i] * scale;
      }
    }
}

// ---- backward: dK, dV (waves own keys) --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                              const float* __restrict__ lse, const float* __restrict__ dout,
                                                              int seq, int heads, float scale, float* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) float Qs[kS][kLd];
  __shared__ __attribute__((aligned(16))) float dOs[kS][kLd];
  __shared__ float l2s[kS], dls[kS];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long long rs = 3LL * heads * kD, os = (long long)heads * kD;
  const float* base = qkv + (long long)b * seq * rs + h * kD;
  const float* dob = dout + (long long)b * seq * os + h * kD;
  const float* ob = out + (long long)b * seq * os + h * kD;
  load_rows(Qs, base, seq, rs);
  load_rows(dOs, dob, seq, os);
  {   // delta[q] = <dO[q], O[q]>, two threads per query
    const int q = threadIdx.x >> 1, half = threadIdx.x & 1;
    float part = 0.f;
    if (q < seq) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(dob + q * os + 32 * half + 4 * u);
        const f32x4 y = *reinterpret_cast<const f32x4*>(ob + q * os + 32 * half + 4 * u);
        const f32x4 pr = x * y;
        part += pr.x + pr.y + pr.z + pr.w;
      }
    }
    part += __shfl_xor(part, 1);
    if (half == 0) {
      dls[q] = part;
      l2s[q] = q < seq ? lse[((long long)b * heads + h) * seq + q] * kLog2e : 1e30f;   // beyond the sequence: p = 0
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int k0 = 32 * wv;
  f32x4 bk[2][4], bv[2][4];
  load_frags(bk, base + heads * kD, k0, seq, rs, scale * kLog2e);
  load_frags(bv, base + 2 * heads * kD, k0, seq, rs, 1.f);
  __syncthreads();
  if (k0 >= seq) return;

  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[kt][dt] = dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int qt = 0; qt < 8; ++qt) {
    if (16 * qt >= seq) break;
    f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 aq = *reinterpret_cast<const f32x4*>(&Qs[16 * qt + c][16 * t + 4 * j]);
      const f32x4 ad = *reinterpret_cast<const f32x4*>(&dOs[16 * qt + c][16 * t + 4 * j]);
      s[0] = mfma_k4(aq, bk[0][t], s[0]);
      s[1] = mfma_k4(aq, bk[1][t], s[1]);
      dp[0] = mfma_k4(ad, bv[0][t], dp[0]);
      dp[1] = mfma_k4(ad, bv[1][t], dp[1]);
    }
    // s[kt][i] = S[query 16 qt + 4 j + i][key k0 + 16 kt + c]: the A operand (row = key c, k = query 4 j + s) below
    f32x4 p[2], ds[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float l = l2s[16 * qt + 4 * j + i], d = dls[16 * qt + 4 * j + i];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        p[kt][i] = exp2f(s[kt][i] - l);
        ds[kt][i] = p[kt][i] * (dp[kt][i] - d);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 bd, bqv;
      bd.x = dOs[16 * qt + 4 * j + 0][16 * dt + c];
      bd.y = dOs[16 * qt + 4 * j + 1][16 * dt + c];
      bd.z = dOs[16 * qt + 4 * j + 2][16 * dt + c];
      bd.w = dOs[16 * qt + 4 * j + 3][16 * dt + c];
      bqv.x = Qs[16 * qt + 4 * j + 0][16 * dt + c];
      bqv.y = Qs[16 * qt + 4 * j + 1][16 * dt + c];
      bqv.z = Qs[16 * qt + 4 * j + 2][16 * dt + c];
      bqv.w = Qs[16 * qt + 4 * j + 3][16 * dt + c];
      dv[0][dt] = mfma_k4(p[0], bd, dv[0][dt]);
      dv[1][dt] = mfma_k4(p[1], bd, dv[1][dt]);
      dk[0][dt] = mfma_k4(ds[0], bqv, dk[0][dt]);
      dk[1][dt] = mfma_k4(ds[1], bqv, dk[1][dt]);
    }
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = k0 + 16 * kt + 4 * j + i;
      if (key < seq) {
        float* dst = dqkv + ((long long)b * seq + key) * rs + h * kD + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dst[heads * kD + 16 * dt] = dk[kt][dt][i] * scale;
          dst[2 * heads * kD + 16 * dt] = dv[kt][dt][i];
        }
      }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace efg

using namespace efg;

// qkv [batch, seq, 3, heads, 64] (the fused in-projection output) -> out [batch, seq, heads, 64], lse [batch, heads, seq]
// (natural-log sum of exp(scale * <q, k>)).  1 <= seq <= 128.
extern "C" int efg_attention_fwd_f32(const float* qkv, int64_t batch, int seq, int heads, float scale, float* out, float* lse,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(batch >= 0 && heads >= 1 && seq >= 1 && seq <= kS, "attention_fwd: 1 <= seq <= %d, head width %d", kS, kD);
  if (batch == 0) return EFG_OK;
  EFG_CHECK_ARG(qkv && out && lse && aligned16(qkv), "attention_fwd: null or unaligned pointer");
  EFG_CHECK_ARG(batch * heads < (1LL << 31), "attention_fwd: too many sequences");
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0, stream, qkv, seq, heads, scale, out, lse);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

// dqkv [batch, seq, 3, heads, 64] = gradient of the in-projection output, every element written.
extern "C" int efg_attention_bwd_f32(const float* qkv, const float* out, const float* lse, const float* dout, int64_t batch, int seq,
                                     int heads, float scale, float* dqkv, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(batch >= 0 && heads >= 1 && seq >= 1 && seq <= kS, "attention_bwd: 1 <= seq <= %d, head width %d", kS, kD);
  if (batch == 0) return EFG_OK;
  EFG_CHECK_ARG(qkv && out && lse && dout && dqkv && aligned16(qkv) && aligned16(out) && aligned16(dout),
                "attention_bwd: null or unaligned pointer");
  EFG_CHECK_ARG(batch * heads < (1LL << 31), "attention_bwd: too many sequences");
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0, stream, qkv, out, lse, dout, seq, heads,
                     scale, dqkv);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0, stream, qkv, out, lse, dout, seq, heads,
                     scale, dqkv);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
