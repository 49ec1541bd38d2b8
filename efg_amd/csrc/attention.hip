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
// through (batch, row) strides, i.e. straight from the fused in-projection output [B, S, 3, H, 64], and dQ, dK, dV are
// written with the same strides into ONE tensor of that layout, the gradient of the projection.  The same kernels serve
// the encoder's cross-attention (its summary token, 1 query, against the 128 points: Q [B, 1, H, 64], K | V
// [B, 128, 2, H, 64]); there only the first wave has queries and the kernels are bound by reading K and V.
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

// 2^x for x <= ~0: the bare v_exp_f32 (1 ulp; below -126 it flushes to 0, which is what a vanishing probability should be);
// exp2f() wraps it in a range test + ldexp for denormal results, 5 instructions per value in the softmax loops
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ f32x4 mfma_k4(const f32x4 a, const f32x4 b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

// rows [0, seq) of two [seq, 64] strided matrices into LDS, rows beyond zero-filled.  All 16 global loads of a thread are
// issued before the first LDS write (unconditional loads from a clamped row, zeroed afterwards): written as a loop of
// load-then-store the compiler waits for every pair of loads, and with two waves per SIMD nothing hides that latency.
__device__ __forceinline__ void load_rows2(float (*dst_a)[kLd], const float* __restrict__ src_a, int seq_a, long long stride_a,
                                           float (*dst_b)[kLd], const float* __restrict__ src_b, int seq_b, long long stride_b) {
  f32x4 va[8], vb[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = threadIdx.x + 256 * it, row = idx >> 4, c4 = idx & 15;
    va[it] = *reinterpret_cast<const f32x4*>(src_a + min(row, seq_a - 1) * stride_a + 4 * c4);
    vb[it] = *reinterpret_cast<const f32x4*>(src_b + min(row, seq_b - 1) * stride_b + 4 * c4);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = threadIdx.x + 256 * it, row = idx >> 4, c4 = idx & 15;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(&dst_a[row][4 * c4]) = row < seq_a ? va[it] : zero;
    *reinterpret_cast<f32x4*>(&dst_b[row][4 * c4]) = row < seq_b ? vb[it] : zero;
  }
}

// B-operand fragments of 2 x 16 rows starting at row0: frag[tile][t] = src[row0 + 16 tile + (lane & 15)][16 t + 4 j ..]
__device__ __forceinline__ void load_frags(f32x4 (&frag)[2][4], const float* __restrict__ src, int row0, int seq, long long stride,
                                           float mul) {
  const int lane = threadIdx.x & 63, j = lane >> 4, c = lane & 15;
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    const int row = row0 + 16 * tile + c;
    const float m = row < seq ? mul : 0.f;   // rows beyond the sequence: a clamped (valid) load, zeroed
#pragma unroll
    for (int t = 0; t < 4; ++t)
      frag[tile][t] = *reinterpret_cast<const f32x4*>(src + min(row, seq - 1) * stride + 16 * t + 4 * j) * m;
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
// Operand addressing shared by the three kernels: element (b, row, h, d) of Q at q + b * q_bs + row * q_rs + h * 64 + d,
// of K / V at k|v + b * kv_bs + row * kv_rs + h * 64 + d; gradients use the strides of the operand they belong to.
struct Operands {
  const float* q;
  const float* k;
  const float* v;
  long long q_bs, q_rs, kv_bs, kv_rs;
  int sq, sk, heads;
  float scale;
};

__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(const Operands a, float* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) float Ks[kS][kLd];
  __shared__ __attribute__((aligned(16))) float Vs[kS][kLd];
  const int heads = a.heads, seq = a.sk, sq = a.sq;
  const float scale = a.scale;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int q0 = 32 * wv;
  f32x4 bq[2][4];
  load_frags(bq, a.q + b * a.q_bs + h * kD, q0, sq, a.q_rs, scale * kLog2e);
  load_rows2(Ks, a.k + b * a.kv_bs + h * kD, seq, a.kv_rs, Vs, a.v + b * a.kv_bs + h * kD, seq, a.kv_rs);
  __syncthreads();
  if (q0 >= sq) return;

  // S^T tiles: acc[kt][qt][i] = log2e * scale * <K[16 kt + 4 j + i], Q[q0 + 16 qt + c]>.  The A fragments of key tile
  // kt + 1 are read while tile kt multiplies (the compiler otherwise waits on every pair of reads right before its MFMAs).
  f32x4 acc[8][2];
  f32x4 ak[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) ak[0][t] = *reinterpret_cast<const f32x4*>(&Ks[c][16 * t + 4 * j]);
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    acc[kt][0] = acc[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (16 * kt < seq) {
      if (kt + 1 < 8) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          ak[(kt + 1) & 1][t] = *reinterpret_cast<const f32x4*>(&Ks[16 * (kt + 1) + c][16 * t + 4 * j]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[kt][0] = mfma_k4(ak[kt & 1][t], bq[0][t], acc[kt][0]);
        acc[kt][1] = mfma_k4(ak[kt & 1][t], bq[1][t], acc[kt][1]);
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
        const float p = fast_exp2(acc[kt][qt][i] - m);
        acc[kt][qt][i] = p;
        sum += p;
      }
    sum = group_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) acc[kt][qt] *= inv;
    const int q = q0 + 16 * qt + c;
    if (j == 0 && q < sq) lse[((long long)b * heads + h) * sq + q] = (m + log2f(sum)) * kLn2;
  }
  // O = P V: the S^T tile is the A operand (row = query c, k = key 4 j + s)
  f32x4 o[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto v_frag = [&](int u) {   // B operand of step u = 4 kt + dt: V[16 kt + 4 j + s][16 dt + c], s = 0..3
    const float* p = &Vs[16 * (u >> 2) + 4 * j][16 * (u & 3) + c];
    return f32x4{p[0], p[kLd], p[2 * kLd], p[3 * kLd]};
  };
  const int steps = 4 * min(8, (seq + 15) >> 4);
  f32x4 vf[2];
  vf[0] = v_frag(0);
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    if (u < steps) {
      if (u + 1 < 32) vf[(u + 1) & 1] = v_frag(u + 1);   // (a tile past `seq` is read but not used: rows exist, zero-filled)
      o[0][u & 3] = mfma_k4(acc[u >> 2][0], vf[u & 1], o[0][u & 3]);
      o[1][u & 3] = mfma_k4(acc[u >> 2][1], vf[u & 1], o[1][u & 3]);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + 16 * qt + 4 * j + i;
      if (q < sq) {
        float* dst = out + (((long long)b * sq + q) * heads + h) * kD + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dst[16 * dt] = o[qt][dt][i];
      }
    }
}

// ---- backward: dQ (waves own queries) ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_kernel(const Operands a, const float* __restrict__ out,
                                                             const float* __restrict__ lse, const float* __restrict__ dout,
                                                             float* __restrict__ dq_out) {
  __shared__ __attribute__((aligned(16))) float Ks[kS][kLd];
  __shared__ __attribute__((aligned(16))) float Vs[kS][kLd];
  const int heads = a.heads, seq = a.sk, sq = a.sq;
  const float scale = a.scale;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long long os = (long long)heads * kD;
  load_rows2(Ks, a.k + b * a.kv_bs + h * kD, seq, a.kv_rs, Vs, a.v + b * a.kv_bs + h * kD, seq, a.kv_rs);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int q0 = 32 * wv;
  f32x4 bq[2][4], bdo[2][4];
  load_frags(bq, a.q + b * a.q_bs + h * kD, q0, sq, a.q_rs, scale * kLog2e);
  load_frags(bdo, dout + (long long)b * sq * os + h * kD, q0, sq, os, 1.f);
  float delta[2], l2[2];
  {
    f32x4 bo[2][4];
    load_frags(bo, out + (long long)b * sq * os + h * kD, q0, sq, os, 1.f);
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
      l2[qt] = q < sq ? lse[((long long)b * heads + h) * sq + q] * kLog2e : 0.f;
    }
  }
  __syncthreads();
  if (q0 >= sq) return;

  f32x4 dq[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per key tile: the A fragments of tile kt + 1 and the K^T fragments of this tile's dQ product are read before the
  // S / dP products start, so the LDS latency sits under 32 MFMAs
  f32x4 ak[2][4], av[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    ak[0][t] = *reinterpret_cast<const f32x4*>(&Ks[c][16 * t + 4 * j]);
    av[0][t] = *reinterpret_cast<const f32x4*>(&Vs[c][16 * t + 4 * j]);
  }
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    if (16 * kt < seq) {
      if (kt + 1 < 8) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ak[(kt + 1) & 1][t] = *reinterpret_cast<const f32x4*>(&Ks[16 * (kt + 1) + c][16 * t + 4 * j]);
          av[(kt + 1) & 1][t] = *reinterpret_cast<const f32x4*>(&Vs[16 * (kt + 1) + c][16 * t + 4 * j]);
        }
      }
      f32x4 kv[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float* p = &Ks[16 * kt + 4 * j][16 * dt + c];
        kv[dt] = f32x4{p[0], p[kLd], p[2 * kLd], p[3 * kLd]};
      }
      f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[0] = mfma_k4(ak[kt & 1][t], bq[0][t], s[0]);
        s[1] = mfma_k4(ak[kt & 1][t], bq[1][t], s[1]);
        dp[0] = mfma_k4(av[kt & 1][t], bdo[0][t], dp[0]);
        dp[1] = mfma_k4(av[kt & 1][t], bdo[1][t], dp[1]);
      }
      f32x4 ds[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = (16 * kt + 4 * j + i < seq) ? fast_exp2(s[qt][i] - l2[qt]) : 0.f;
          ds[qt][i] = p * (dp[qt][i] - delta[qt]);
        }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dq[0][dt] = mfma_k4(ds[0], kv[dt], dq[0][dt]);
        dq[1][dt] = mfma_k4(ds[1], kv[dt], dq[1][dt]);
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + 16 * qt + 4 * j + i;
      if (q < sq) {
        float* dst = dq_out + b * a.q_bs + q * a.q_rs + h * kD + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dst[16 * dt] = dq[qt][dt][i] * scale;
      }
    }
}

// ---- backward: dK, dV (waves own keys) --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) attn_bwd_dkv_kernel(const Operands a, const float* __restrict__ out,
                                                              const float* __restrict__ lse, const float* __restrict__ dout,
                                                              float* __restrict__ dk_out, float* __restrict__ dv_out) {
  __shared__ __attribute__((aligned(16))) float Qs[kS][kLd];
  __shared__ __attribute__((aligned(16))) float dOs[kS][kLd];
  __shared__ float l2s[kS], dls[kS];
  const int heads = a.heads, seq = a.sk, sq = a.sq;
  const float scale = a.scale;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long long os = (long long)heads * kD;
  const float* dob = dout + (long long)b * sq * os + h * kD;
  const float* ob = out + (long long)b * sq * os + h * kD;
  load_rows2(Qs, a.q + b * a.q_bs + h * kD, sq, a.q_rs, dOs, dob, sq, os);
  {   // delta[q] = <dO[q], O[q]>, two threads per query
    const int q = threadIdx.x >> 1, half = threadIdx.x & 1;
    float part = 0.f;
    if (q < sq) {
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
      l2s[q] = q < sq ? lse[((long long)b * heads + h) * sq + q] * kLog2e : 1e30f;   // beyond the sequence: p = 0
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int k0 = 32 * wv;
  f32x4 bk[2][4], bv[2][4];
  load_frags(bk, a.k + b * a.kv_bs + h * kD, k0, seq, a.kv_rs, scale * kLog2e);
  load_frags(bv, a.v + b * a.kv_bs + h * kD, k0, seq, a.kv_rs, 1.f);
  __syncthreads();
  if (k0 >= seq) return;

  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[kt][dt] = dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 aq[2][4], ad[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    aq[0][t] = *reinterpret_cast<const f32x4*>(&Qs[c][16 * t + 4 * j]);
    ad[0][t] = *reinterpret_cast<const f32x4*>(&dOs[c][16 * t + 4 * j]);
  }
#pragma unroll
  for (int qt = 0; qt < 8; ++qt) {
    if (16 * qt < sq) {
      if (qt + 1 < 8) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          aq[(qt + 1) & 1][t] = *reinterpret_cast<const f32x4*>(&Qs[16 * (qt + 1) + c][16 * t + 4 * j]);
          ad[(qt + 1) & 1][t] = *reinterpret_cast<const f32x4*>(&dOs[16 * (qt + 1) + c][16 * t + 4 * j]);
        }
      }
      f32x4 bd[4], bqv[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float* pd = &dOs[16 * qt + 4 * j][16 * dt + c];
        const float* pq = &Qs[16 * qt + 4 * j][16 * dt + c];
        bd[dt] = f32x4{pd[0], pd[kLd], pd[2 * kLd], pd[3 * kLd]};
        bqv[dt] = f32x4{pq[0], pq[kLd], pq[2 * kLd], pq[3 * kLd]};
      }
      f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[0] = mfma_k4(aq[qt & 1][t], bk[0][t], s[0]);
        s[1] = mfma_k4(aq[qt & 1][t], bk[1][t], s[1]);
        dp[0] = mfma_k4(ad[qt & 1][t], bv[0][t], dp[0]);
        dp[1] = mfma_k4(ad[qt & 1][t], bv[1][t], dp[1]);
      }
      // s[kt][i] = S[query 16 qt + 4 j + i][key k0 + 16 kt + c]: the A operand (row = key c, k = query 4 j + s) below
      f32x4 p[2], ds[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float l = l2s[16 * qt + 4 * j + i], d = dls[16 * qt + 4 * j + i];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          p[kt][i] = fast_exp2(s[kt][i] - l);
          ds[kt][i] = p[kt][i] * (dp[kt][i] - d);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[0][dt] = mfma_k4(p[0], bd[dt], dv[0][dt]);
        dv[1][dt] = mfma_k4(p[1], bd[dt], dv[1][dt]);
        dk[0][dt] = mfma_k4(ds[0], bqv[dt], dk[0][dt]);
        dk[1][dt] = mfma_k4(ds[1], bqv[dt], dk[1][dt]);
      }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = k0 + 16 * kt + 4 * j + i;
      if (key < seq) {
        const long long at = b * a.kv_bs + key * a.kv_rs + h * kD + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dk_out[at + 16 * dt] = dk[kt][dt][i] * scale;
          dv_out[at + 16 * dt] = dv[kt][dt][i];
        }
      }
    }
}

// ==== long sequences, 32-wide heads (ConQueR's decoder self-attention: 2 x 8 heads x 1240 queries, bool mask) ============
// Same S^T trick, blocked over 128 keys with the online softmax: a workgroup owns 64 queries (a wave 16 = one tile) of one
// (sequence, head) and walks the key blocks; the running max / sum of query c live in the lanes of column c, the O tile has
// queries on its rows, so the per-block rescale factors move with one __shfl per row.  The boolean attention mask comes
// bit-packed ([S, ceil(S / 32)] words, bit = 1: not allowed): a lane reads 4 words per 128-key block.
constexpr int kDL = 32;    // head width
constexpr int kLdL = 36;   // LDS row stride (floats)
constexpr int kBlk = 128;  // keys (or queries, in the dK / dV kernel) per LDS block

struct LongOperands {
  const float* q;
  const float* k;
  const float* v;
  long long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs;
  const unsigned* mask;   // [s, mask_words] or null
  int mask_words;
  int s, heads;
  float scale;
};

// rows [row0, row0 + 128) of two [s, 32] strided matrices: global -> registers (`fetch`, clamped rows) and registers -> LDS
// (`store`, zero beyond s).  The kernels fetch block n + 1 right after storing block n, so the global latency runs under
// block n's MFMAs: at ConQueR's shape there are only ~1.25 workgroups per CU and nothing else would hide it.
struct BlockRegs {
  f32x4 a[4], b[4];
};
__device__ __forceinline__ void fetch_block2(BlockRegs& r, const float* __restrict__ src_a, long long stride_a,
                                             const float* __restrict__ src_b, long long stride_b, int row0, int s) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = threadIdx.x + 256 * it, row = min(row0 + (idx >> 3), s - 1), c4 = idx & 7;
    r.a[it] = *reinterpret_cast<const f32x4*>(src_a + row * stride_a + 4 * c4);
    r.b[it] = *reinterpret_cast<const f32x4*>(src_b + row * stride_b + 4 * c4);
  }
}
__device__ __forceinline__ void store_block2(float (*dst_a)[kLdL], float (*dst_b)[kLdL], const BlockRegs& r, int row0, int s) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = threadIdx.x + 256 * it, rr = idx >> 3, c4 = idx & 7;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(&dst_a[rr][4 * c4]) = row0 + rr < s ? r.a[it] : zero;
    *reinterpret_cast<f32x4*>(&dst_b[rr][4 * c4]) = row0 + rr < s ? r.b[it] : zero;
  }
}

// B-operand fragments of 16 rows: frag[t] = src[row0 + (lane & 15)][16 t + 4 j ..] * mul, rows beyond s zero
__device__ __forceinline__ void load_frag16(f32x4 (&frag)[2], const float* __restrict__ src, int row0, int s, long long stride,
                                            float mul) {
  const int lane = threadIdx.x & 63, j = lane >> 4, c = lane & 15;
  const int row = row0 + c;
  const float m = row < s ? mul : 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    frag[t] = *reinterpret_cast<const f32x4*>(src + min(row, s - 1) * stride + 16 * t + 4 * j) * m;
}

__device__ __forceinline__ f32x4 col_frag(const float (*m)[kLdL], int row, int col) {   // rows row .. row + 3 of one column
  return f32x4{m[row][col], m[row + 1][col], m[row + 2][col], m[row + 3][col]};
}

__global__ void __launch_bounds__(256) attn_long_fwd_kernel(const LongOperands a, float* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) float Ks[kBlk][kLdL];
  __shared__ __attribute__((aligned(16))) float Vs[kBlk][kLdL];
  const int s = a.s, heads = a.heads;
  const int b = blockIdx.y / heads, h = blockIdx.y % heads;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int q0 = 64 * blockIdx.x + 16 * wv;
  const float* kp = a.k + b * a.k_bs + h * kDL;
  const float* vp = a.v + b * a.v_bs + h * kDL;
  f32x4 bq[2];
  load_frag16(bq, a.q + b * a.q_bs + h * kDL, q0, s, a.q_rs, a.scale * kLog2e);
  const unsigned* mrow = a.mask ? a.mask + (long long)min(q0 + c, s - 1) * a.mask_words : nullptr;
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  BlockRegs nxt;
  unsigned mw_nxt[4] = {0u, 0u, 0u, 0u};
  auto fetch = [&](int k0) {
    fetch_block2(nxt, kp, a.k_rs, vp, a.v_rs, k0, s);
    if (mrow) {
#pragma unroll
      for (int w = 0; w < 4; ++w) mw_nxt[w] = mrow[min((k0 >> 5) + w, a.mask_words - 1)];   // (words past the row: keys >= s)
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < s; k0 += kBlk) {
    __syncthreads();
    store_block2(Ks, Vs, nxt, k0, s);
    unsigned mw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) mw[w] = mw_nxt[w];
    if (k0 + kBlk < s) fetch(k0 + kBlk);
    __syncthreads();
    f32x4 acc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
        acc[kt] = mfma_k4(*reinterpret_cast<const f32x4*>(&Ks[16 * kt + c][16 * t + 4 * j]), bq[t], acc[kt]);
    }
    // a block inside the sequence with no masked pair for any query of the wave skips the per-element tests
    const bool plain = k0 + kBlk <= s && __builtin_amdgcn_ballot_w64((mw[0] | mw[1] | mw[2] | mw[3]) != 0u) == 0ull;
    float bm = -INFINITY;
    if (plain) {
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) bm = fmaxf(bm, fmaxf(fmaxf(acc[kt][0], acc[kt][1]), fmaxf(acc[kt][2], acc[kt][3])));
    } else {
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = 16 * kt + 4 * j + i;
          const bool dead = k0 + key >= s || ((mw[kt >> 1] >> (key & 31)) & 1u);
          acc[kt][i] = dead ? -INFINITY : acc[kt][i];
          bm = fmaxf(bm, acc[kt][i]);
        }
    }
    const float m_new = fmaxf(m_run, group_max(bm));
    const float m_use = m_new == -INFINITY ? 0.f : m_new;   // nothing allowed so far: every p below is exp2(-inf) = 0
    float bs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[kt][i] = fast_exp2(acc[kt][i] - m_use);
        bs += acc[kt][i];
      }
    const float alpha = fast_exp2(m_run - m_use);   // m_run = -inf -> 0
    l_run = l_run * alpha + group_sum(bs);
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ai = __shfl(alpha, 4 * j + i);   // lane 4 j + i holds the factor of query 4 j + i
      o[0][i] *= ai;
      o[1][i] *= ai;
    }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      o[0] = mfma_k4(acc[kt], col_frag(Vs, 16 * kt + 4 * j, c), o[0]);
      o[1] = mfma_k4(acc[kt], col_frag(Vs, 16 * kt + 4 * j, 16 + c), o[1]);
    }
  }
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;   // a query with no allowed key: zeros (PyTorch: NaN)
  if (j == 0 && q0 + c < s) lse[((long long)b * heads + h) * s + q0 + c] = (m_run + log2f(l_run)) * kLn2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float ii = __shfl(inv, 4 * j + i);
    const int q = q0 + 4 * j + i;
    if (q < s) {
      float* dst = out + (((long long)b * s + q) * heads + h) * kDL + c;
      dst[0] = o[0][i] * ii;
      dst[16] = o[1][i] * ii;
    }
  }
}

__global__ void __launch_bounds__(256) attn_long_bwd_dq_kernel(const LongOperands a, const float* __restrict__ out,
                                                               const float* __restrict__ lse, const float* __restrict__ dout,
                                                               float* __restrict__ dq_out, float* __restrict__ delta_ws) {
  __shared__ __attribute__((aligned(16))) float Ks[kBlk][kLdL];
  __shared__ __attribute__((aligned(16))) float Vs[kBlk][kLdL];
  const int s = a.s, heads = a.heads;
  const int b = blockIdx.y / heads, h = blockIdx.y % heads;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int q0 = 64 * blockIdx.x + 16 * wv;
  const long long os = (long long)heads * kDL;
  const float* kp = a.k + b * a.k_bs + h * kDL;
  const float* vp = a.v + b * a.v_bs + h * kDL;
  f32x4 bq[2], bdo[2], bo[2];
  load_frag16(bq, a.q + b * a.q_bs + h * kDL, q0, s, a.q_rs, a.scale * kLog2e);
  load_frag16(bdo, dout + (long long)b * s * os + h * kDL, q0, s, os, 1.f);
  load_frag16(bo, out + (long long)b * s * os + h * kDL, q0, s, os, 1.f);
  float part = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 pr = bo[t] * bdo[t];
    part += pr.x + pr.y + pr.z + pr.w;
  }
  const float delta = group_sum(part);
  const float l2 = q0 + c < s ? lse[((long long)b * heads + h) * s + q0 + c] * kLog2e : 0.f;
  const unsigned* mrow = a.mask ? a.mask + (long long)min(q0 + c, s - 1) * a.mask_words : nullptr;
  f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (j == 0 && q0 + c < s) delta_ws[((long long)b * heads + h) * s + q0 + c] = delta;   // for the dK / dV kernel
  BlockRegs nxt;
  unsigned mw_nxt[4] = {0u, 0u, 0u, 0u};
  auto fetch = [&](int k0) {
    fetch_block2(nxt, kp, a.k_rs, vp, a.v_rs, k0, s);
    if (mrow) {
#pragma unroll
      for (int w = 0; w < 4; ++w) mw_nxt[w] = mrow[min((k0 >> 5) + w, a.mask_words - 1)];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < s; k0 += kBlk) {
    __syncthreads();
    store_block2(Ks, Vs, nxt, k0, s);
    unsigned mw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) mw[w] = mw_nxt[w];
    if (k0 + kBlk < s) fetch(k0 + kBlk);
    const bool plain = k0 + kBlk <= s && __builtin_amdgcn_ballot_w64((mw[0] | mw[1] | mw[2] | mw[3]) != 0u) == 0ull;
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sv = mfma_k4(*reinterpret_cast<const f32x4*>(&Ks[16 * kt + c][16 * t + 4 * j]), bq[t], sv);
        dp = mfma_k4(*reinterpret_cast<const f32x4*>(&Vs[16 * kt + c][16 * t + 4 * j]), bdo[t], dp);
      }
      f32x4 ds;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = 16 * kt + 4 * j + i;
        const bool dead = !plain && (k0 + key >= s || ((mw[kt >> 1] >> (key & 31)) & 1u));
        const float p = dead ? 0.f : fast_exp2(sv[i] - l2);
        ds[i] = p * (dp[i] - delta);
      }
      dq[0] = mfma_k4(ds, col_frag(Ks, 16 * kt + 4 * j, c), dq[0]);
      dq[1] = mfma_k4(ds, col_frag(Ks, 16 * kt + 4 * j, 16 + c), dq[1]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = q0 + 4 * j + i;
    if (q < s) {
      float* dst = dq_out + b * a.q_bs + q * a.q_rs + h * kDL + c;
      dst[0] = dq[0][i] * a.scale;
      dst[16] = dq[1][i] * a.scale;
    }
  }
}

__global__ void __launch_bounds__(256) attn_long_bwd_dkv_kernel(const LongOperands a, const float* __restrict__ delta_ws,
                                                                const float* __restrict__ lse, const float* __restrict__ dout,
                                                                float* __restrict__ dk_out, float* __restrict__ dv_out) {
  __shared__ __attribute__((aligned(16))) float Qs[kBlk][kLdL];
  __shared__ __attribute__((aligned(16))) float dOs[kBlk][kLdL];
  __shared__ float l2s[kBlk], dls[kBlk];
  __shared__ unsigned msk[4][kBlk];
  const int s = a.s, heads = a.heads;
  const int b = blockIdx.y / heads, h = blockIdx.y % heads;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane >> 4, c = lane & 15;
  const int k0 = 64 * blockIdx.x + 16 * wv;       // this wave's 16 keys: inside one mask word
  const long long os = (long long)heads * kDL;
  const float* qp = a.q + b * a.q_bs + h * kDL;
  const float* dob = dout + (long long)b * s * os + h * kDL;
  f32x4 bk[2], bv[2];
  load_frag16(bk, a.k + b * a.k_bs + h * kDL, k0, s, a.k_rs, a.scale * kLog2e);
  load_frag16(bv, a.v + b * a.v_bs + h * kDL, k0, s, a.v_rs, 1.f);
  const int kbit = (k0 & 31) + c;                 // bit of key k0 + c in its word
  f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  BlockRegs nxt;
  float nl = 0.f, nd = 0.f;      // threads 0..127: log-sum-exp and delta of query r0 + threadIdx.x
  unsigned nm[2] = {0u, 0u};     // this wave's mask word of queries r0 + lane, r0 + lane + 64
  const long long stat0 = ((long long)b * heads + h) * s;
  auto fetch = [&](int r0) {
    fetch_block2(nxt, qp, a.q_rs, dob, os, r0, s);
    if (threadIdx.x < kBlk) {
      const int q = min(r0 + (int)threadIdx.x, s - 1);
      nl = lse[stat0 + q];
      nd = delta_ws[stat0 + q];
    }
    if (a.mask) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        nm[u] = a.mask[(long long)min(r0 + lane + 64 * u, s - 1) * a.mask_words + (min(k0, s - 1) >> 5)];
    }
  };
  fetch(0);
  for (int r0 = 0; r0 < s; r0 += kBlk) {
    __syncthreads();
    store_block2(Qs, dOs, nxt, r0, s);
    if (threadIdx.x < kBlk) {
      const bool live = r0 + (int)threadIdx.x < s;
      l2s[threadIdx.x] = live ? nl * kLog2e : 1e30f;   // beyond the sequence: p = 0
      dls[threadIdx.x] = live ? nd : 0.f;
    }
    msk[wv][lane] = nm[0];
    msk[wv][lane + 64] = nm[1];
    if (r0 + kBlk < s) fetch(r0 + kBlk);
    __syncthreads();
    if (k0 < s) {
#pragma unroll 2
      for (int qt = 0; qt < 8; ++qt) {
        if (r0 + 16 * qt >= s) break;
        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          sv = mfma_k4(*reinterpret_cast<const f32x4*>(&Qs[16 * qt + c][16 * t + 4 * j]), bk[t], sv);
          dp = mfma_k4(*reinterpret_cast<const f32x4*>(&dOs[16 * qt + c][16 * t + 4 * j]), bv[t], dp);
        }
        f32x4 p, ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 16 * qt + 4 * j + i;
          const bool dead = (msk[wv][r] >> kbit) & 1u;
          p[i] = dead ? 0.f : fast_exp2(sv[i] - l2s[r]);
          ds[i] = p[i] * (dp[i] - dls[r]);
        }
        dv[0] = mfma_k4(p, col_frag(dOs, 16 * qt + 4 * j, c), dv[0]);
        dv[1] = mfma_k4(p, col_frag(dOs, 16 * qt + 4 * j, 16 + c), dv[1]);
        dk[0] = mfma_k4(ds, col_frag(Qs, 16 * qt + 4 * j, c), dk[0]);
        dk[1] = mfma_k4(ds, col_frag(Qs, 16 * qt + 4 * j, 16 + c), dk[1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = k0 + 4 * j + i;
    if (key < s) {
      float* kd = dk_out + b * a.k_bs + key * a.k_rs + h * kDL + c;
      float* vd = dv_out + b * a.v_bs + key * a.v_rs + h * kDL + c;
      kd[0] = dk[0][i] * a.scale;
      kd[16] = dk[1][i] * a.scale;
      vd[0] = dv[0][i];
      vd[16] = dv[1][i];
    }
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace efg

using namespace efg;

static int check_operands(const char* who, const float* q, int64_t q_bs, int64_t q_rs, const float* k, const float* v, int64_t kv_bs,
                          int64_t kv_rs, int64_t batch, int seq_q, int seq_k, int heads) {
  EFG_CHECK_ARG(batch >= 0 && heads >= 1 && seq_q >= 1 && seq_q <= kS && seq_k >= 1 && seq_k <= kS,
                "%s: 1 <= seq_q, seq_k <= %d, head width %d", who, kS, kD);
  EFG_CHECK_ARG(batch * heads < (1LL << 31), "%s: too many sequences", who);
  if (batch == 0) return EFG_OK;
  EFG_CHECK_ARG(q && k && v && aligned16(q) && aligned16(k) && aligned16(v), "%s: null or unaligned operand", who);
  EFG_CHECK_ARG(q_bs % 4 == 0 && q_rs % 4 == 0 && kv_bs % 4 == 0 && kv_rs % 4 == 0 && q_rs >= (int64_t)heads * kD &&
                    kv_rs >= (int64_t)heads * kD,
                "%s: strides must be multiples of 4 floats and rows at least heads x %d wide", who, kD);
  return EFG_OK;
}

// softmax(scale * Q K^T) V per (sequence, head).  Element (b, row, h, d) of Q at q + b * q_batch_stride + row * q_row_stride
// + h * 64 + d, of K / V at k|v + b * kv_batch_stride + row * kv_row_stride + h * 64 + d (strides in floats), which
// covers the fused in-projection layouts [B, S, 3, H, 64] (self-attention) and [B, Sk, 2, H, 64] + [B, Sq, H, 64] (cross).
// out [batch, seq_q, heads, 64], lse [batch, heads, seq_q].
extern "C" int efg_attention_fwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k, const float* v,
                                     int64_t kv_batch_stride, int64_t kv_row_stride, int64_t batch, int seq_q, int seq_k, int heads,
                                     float scale, float* out, float* lse, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = check_operands("attention_fwd", q, q_batch_stride, q_row_stride, k, v, kv_batch_stride, kv_row_stride, batch, seq_q,
                                seq_k, heads);
  if (rc != EFG_OK || batch == 0) return rc;
  EFG_CHECK_ARG(out && lse, "attention_fwd: null output");
  const Operands a{q, k, v, q_batch_stride, q_row_stride, kv_batch_stride, kv_row_stride, seq_q, seq_k, heads, scale};
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0, stream, a, out, lse);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

// dq / dk / dv are addressed like q / k / v (same strides); every element of the (b, row < seq, h, d) sub-arrays is written.
extern "C" int efg_attention_bwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k, const float* v,
                                     int64_t kv_batch_stride, int64_t kv_row_stride, const float* out, const float* lse,
                                     const float* dout, int64_t batch, int seq_q, int seq_k, int heads, float scale, float* dq,
                                     float* dk, float* dv, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = check_operands("attention_bwd", q, q_batch_stride, q_row_stride, k, v, kv_batch_stride, kv_row_stride, batch, seq_q,
                                seq_k, heads);
  if (rc != EFG_OK || batch == 0) return rc;
  EFG_CHECK_ARG(out && lse && dout && dq && dk && dv && aligned16(out) && aligned16(dout), "attention_bwd: null or unaligned pointer");
  const Operands a{q, k, v, q_batch_stride, q_row_stride, kv_batch_stride, kv_row_stride, seq_q, seq_k, heads, scale};
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0, stream, a, out, lse, dout, dq);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0, stream, a, out, lse, dout, dk, dv);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

// ---- long sequences (any s >= 1), 32-wide heads, optional boolean mask ---------------------------------------------------------
// q / k / v element (b, row, h, d) at ptr + b * batch_stride + row * row_stride + h * 32 + d (strides in floats, multiples of 4).
// mask_bits: u32 [s, mask_words], bit (key & 31) of word key >> 5 of row `query` set = that key is NOT attended (torch's boolean
// attn_mask), null = no mask; the same mask for every sequence and head.  out [batch, s, heads, 32], lse [batch, heads, s].
static int check_long(const char* who, const float* q, const float* k, const float* v, const int64_t* st, int64_t batch, int s,
                      int heads, const uint32_t* mask_bits, int mask_words) {
  EFG_CHECK_ARG(batch >= 0 && heads >= 1 && s >= 1, "%s: bad sizes", who);
  EFG_CHECK_ARG(batch * heads < 65536, "%s: batch x heads must be < 65536", who);
  if (batch == 0) return EFG_OK;
  EFG_CHECK_ARG(q && k && v && aligned16(q) && aligned16(k) && aligned16(v), "%s: null or unaligned operand", who);
  for (int i = 0; i < 6; ++i) EFG_CHECK_ARG(st[i] % 4 == 0, "%s: strides must be multiples of 4 floats", who);
  EFG_CHECK_ARG(!mask_bits || mask_words >= (s + 31) / 32, "%s: mask rows need ceil(s / 32) words", who);
  return EFG_OK;
}

extern "C" int efg_attention_long_fwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k,
                                          int64_t k_batch_stride, int64_t k_row_stride, const float* v, int64_t v_batch_stride,
                                          int64_t v_row_stride, const uint32_t* mask_bits, int mask_words, int64_t batch, int s,
                                          int heads, float scale, float* out, float* lse, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t st[6] = {q_batch_stride, q_row_stride, k_batch_stride, k_row_stride, v_batch_stride, v_row_stride};
  const int rc = check_long("attention_long_fwd", q, k, v, st, batch, s, heads, mask_bits, mask_words);
  if (rc != EFG_OK || batch == 0) return rc;
  EFG_CHECK_ARG(out && lse, "attention_long_fwd: null output");
  const LongOperands a{q, k, v, st[0], st[1], st[2], st[3], st[4], st[5], mask_bits, mask_words, s, heads, scale};
  hipLaunchKernelGGL(attn_long_fwd_kernel, dim3((unsigned)((s + 63) / 64), (unsigned)(batch * heads)), dim3(256), 0, stream, a, out,
                     lse);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_attention_long_bwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k,
                                          int64_t k_batch_stride, int64_t k_row_stride, const float* v, int64_t v_batch_stride,
                                          int64_t v_row_stride, const uint32_t* mask_bits, int mask_words, const float* out,
                                          const float* lse, const float* dout, int64_t batch, int s, int heads, float scale,
                                          float* dq, float* dk, float* dv, float* delta_ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t st[6] = {q_batch_stride, q_row_stride, k_batch_stride, k_row_stride, v_batch_stride, v_row_stride};
  const int rc = check_long("attention_long_bwd", q, k, v, st, batch, s, heads, mask_bits, mask_words);
  if (rc != EFG_OK || batch == 0) return rc;
  EFG_CHECK_ARG(out && lse && dout && dq && dk && dv && delta_ws && aligned16(out) && aligned16(dout),
                "attention_long_bwd: null or unaligned pointer");
  const LongOperands a{q, k, v, st[0], st[1], st[2], st[3], st[4], st[5], mask_bits, mask_words, s, heads, scale};
  const dim3 grid((unsigned)((s + 63) / 64), (unsigned)(batch * heads));
  hipLaunchKernelGGL(attn_long_bwd_dq_kernel, grid, dim3(256), 0, stream, a, out, lse, dout, dq, delta_ws);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_long_bwd_dkv_kernel, grid, dim3(256), 0, stream, a, (const float*)delta_ws, lse, dout, dk, dv);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
