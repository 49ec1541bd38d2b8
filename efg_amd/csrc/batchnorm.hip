// BatchNorm1d (+ residual) (+ ReLU) over sparse-tensor features [M, C] for gfx950, training mode
// (and, further down, GroupNorm over the channels-last BEV map with the same chunked statistics).
//
// The sparse backbone normalises after every convolution (sparse_net.py:85-95,120-165): BatchNorm1d over the M
// active sites, then ReLU, in the residual blocks `relu(bn(conv) + shortcut)`.  PyTorch runs that as 4 kernels
// forward (statistics, running-stat update, transform, ReLU; + 1 for the residual add) and 3 backward, and its
// channels-last statistics kernels reach 0.4 TB/s on these [M, 16..256] tensors.  Here:
//   forward : bn_stats_kernel  -- per-workgroup (count, mean, M2) of a contiguous row chunk (two passes over the
//                                 chunk: the second one hits L2)
//             bn_stats_merge   -- one wave per channel merges the chunks with Chan's formula and updates the running
//                                 statistics
//             bn_apply_kernel  -- y = relu((x - mean) * invstd * w + b + residual), float4 streaming
//   backward: bn_bwd_reduce_kernel + bn_bwd_merge_kernel (sums of dy' and dy' * xhat with dy' = dy * [y > 0])
//             bn_bwd_apply_kernel  (dx, and d residual = dy')
// Lanes run over channels (float4) so every load is a coalesced row segment; reductions are deterministic (fixed
// chunking and merge order).
#include "common.h"

#include <algorithm>

namespace efg {
namespace {

constexpr int kBnBlocks = 1024;  // at most this many row chunks (partials), merged by one wave per channel

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct Chunk {
  long long row_lo, row_hi;
};
__device__ __forceinline__ Chunk chunk_of(long long m, int nblocks, int b) {
  const long long per = (m + nblocks - 1) / nblocks;
  Chunk c;
  c.row_lo = min((long long)b * per, m);
  c.row_hi = min(c.row_lo + per, m);
  return c;
}

// Sum over the block's threads that own the same channel quad (tid % q4); smem: float4[256].  Result valid in the
// threads with tid < q4.
__device__ __forceinline__ float4 quad_sum(float4 v, int q4, float4* smem) {
  smem[threadIdx.x] = v;
  __syncthreads();
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((int)threadIdx.x < q4) {
    for (int t = threadIdx.x; t < 256; t += q4) {
      const float4 o = smem[t];
      r.x += o.x;
      r.y += o.y;
      r.z += o.z;
      r.w += o.w;
    }
  }
  __syncthreads();
  return r;
}

// partial[b][0][c] = count (as float), [1] = mean, [2] = M2 of chunk b
__global__ void __launch_bounds__(256)
bn_stats_kernel(const float* __restrict__ x, long long m, int c, float* __restrict__ partial) {
  __shared__ float4 smem[256];
  const int q4 = c / 4, rstep = 256 / q4;
  const int cq = threadIdx.x % q4, r0 = threadIdx.x / q4;
  const bool lane_on = r0 < rstep;  // 256 % q4 may leave idle threads
  // blockIdx.y = sample for the per-sample statistics of GroupNorm (m rows each); BatchNorm launches one
  x += (long long)blockIdx.y * m * c;
  partial += (long long)blockIdx.y * gridDim.x * 3 * c;
  const Chunk ck = chunk_of(m, gridDim.x, blockIdx.x);
  const float n = (float)(ck.row_hi - ck.row_lo);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane_on)
    for (long long r = ck.row_lo + r0; r < ck.row_hi; r += rstep) {
      const float4 v = ld4(x + r * c + cq * 4);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  float4 tot = quad_sum(s, q4, smem);
  // broadcast the chunk mean back to the owners of the quad
  if ((int)threadIdx.x < q4) smem[threadIdx.x] = make_float4(tot.x / fmaxf(n, 1.f), tot.y / fmaxf(n, 1.f), tot.z / fmaxf(n, 1.f), tot.w / fmaxf(n, 1.f));
  __syncthreads();
  const float4 mu = smem[cq];
  __syncthreads();
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane_on)
    for (long long r = ck.row_lo + r0; r < ck.row_hi; r += rstep) {
      const float4 v = ld4(x + r * c + cq * 4);
      q.x += (v.x - mu.x) * (v.x - mu.x);
      q.y += (v.y - mu.y) * (v.y - mu.y);
      q.z += (v.z - mu.z) * (v.z - mu.z);
      q.w += (v.w - mu.w) * (v.w - mu.w);
    }
  const float4 m2 = quad_sum(q, q4, smem);
  if ((int)threadIdx.x < q4) {
    float* p = partial + (long long)blockIdx.x * 3 * c + cq * 4;
    st4(p, make_float4(n, n, n, n));
    st4(p + c, mu);
    st4(p + 2 * c, m2);
  }
}

// one wave per channel, lanes over the chunks: every lane folds its chunks in order (Chan et al.), then the 64 lane
// triples are combined by a butterfly of pairwise merges
__global__ void __launch_bounds__(64)
bn_stats_merge_kernel(const float* __restrict__ partial, int nchunks, int c, float eps, float momentum,
                      float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ running_mean,
                      float* __restrict__ running_var, long long* __restrict__ num_batches) {
  const int lane = threadIdx.x, ch = blockIdx.x;
  float cnt = 0.f, mean = 0.f, M2 = 0.f;
  // (the chunks of a lane are loaded four at a time and folded in the same order as before: a loop that loads, tests
  // and folds one chunk per trip pays one memory round trip per chunk)
  for (int b0 = lane; b0 < nchunks; b0 += 64 * 4) {
    float nbv[4], mbv[4], qbv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + 64 * u;
      const float* p = partial + (long long)min(b, nchunks - 1) * 3 * c + ch;
      nbv[u] = b < nchunks ? p[0] : 0.f;
      mbv[u] = p[c];
      qbv[u] = p[2 * c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float nb = nbv[u], mb = mbv[u], qb = qbv[u];
      if (nb > 0.f) {
        const float tot_n = cnt + nb, d = mb - mean;
        mean += d * (nb / tot_n);
        M2 += qb + d * d * (cnt * nb / tot_n);
        cnt = tot_n;
      }
    }
  }
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const float n2 = __shfl_xor(cnt, dlt, 64), m2 = __shfl_xor(mean, dlt, 64), q2 = __shfl_xor(M2, dlt, 64);
    const bool low = !(lane & dlt);  // the lower lane of a pair is always the left operand: both compute the same
    const float na = low ? cnt : n2, ma = low ? mean : m2, qa = low ? M2 : q2;
    const float nb = low ? n2 : cnt, mb = low ? m2 : mean, qb = low ? q2 : M2;
    const float tot_n = na + nb;
    if (tot_n > 0.f) {
      const float d = mb - ma;
      mean = ma + d * (nb / tot_n);
      M2 = qa + qb + d * d * (na * nb / tot_n);
    } else {
      mean = 0.f;
      M2 = 0.f;
    }
    cnt = tot_n;
  }
  if (lane == 0) {
    const float var = cnt > 0.f ? M2 / cnt : 0.f;
    mean_out[ch] = mean;
    invstd_out[ch] = 1.0f / sqrtf(var + eps);
    if (running_mean) {
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (cnt > 1.f ? M2 / (cnt - 1.f) : var);
    }
    if (ch == 0 && num_batches) *num_batches += 1;
  }
}

__global__ void __launch_bounds__(256)
bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ w,
                const float* __restrict__ b, const float* __restrict__ mean, const float* __restrict__ invstd,
                long long total4, int c, int relu, float* __restrict__ y) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int ch = (int)((e * 4) % c);
  const float4 v = ld4(x + e * 4), mu = ld4(mean + ch), is = ld4(invstd + ch), ww = ld4(w + ch), bb = ld4(b + ch);
  float4 o;
  o.x = (v.x - mu.x) * is.x * ww.x + bb.x;
  o.y = (v.y - mu.y) * is.y * ww.y + bb.y;
  o.z = (v.z - mu.z) * is.z * ww.z + bb.z;
  o.w = (v.w - mu.w) * is.w * ww.w + bb.w;
  if (res) {
    const float4 r = ld4(res + e * 4);
    o.x += r.x;
    o.y += r.y;
    o.z += r.z;
    o.w += r.w;
  }
  if (relu) {
    o.x = fmaxf(o.x, 0.f);
    o.y = fmaxf(o.y, 0.f);
    o.z = fmaxf(o.z, 0.f);
    o.w = fmaxf(o.w, 0.f);
  }
  st4(y + e * 4, o);
}

// partial[b][0][c] = sum dy', [1][c] = sum dy' * xhat; merged -> dbias[c], dweight[c]
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                     const float* __restrict__ mean, const float* __restrict__ invstd, long long m, int c, int relu,
                     float* __restrict__ partial) {
  __shared__ float4 smem[256];
  const int q4 = c / 4, rstep = 256 / q4;
  const int cq = threadIdx.x % q4, r0 = threadIdx.x / q4;
  const bool lane_on = r0 < rstep;
  const Chunk ck = chunk_of(m, gridDim.x, blockIdx.x);
  const float4 mu = ld4(mean + cq * 4), is = ld4(invstd + cq * 4);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (lane_on)
    for (long long r = ck.row_lo + r0; r < ck.row_hi; r += rstep) {
      const long long o = r * c + cq * 4;
      float4 d = ld4(dy + o);
      const float4 v = ld4(x + o);
      if (relu) {
        const float4 yy = ld4(y + o);
        d.x = yy.x > 0.f ? d.x : 0.f;
        d.y = yy.y > 0.f ? d.y : 0.f;
        d.z = yy.z > 0.f ? d.z : 0.f;
        d.w = yy.w > 0.f ? d.w : 0.f;
      }
      s1.x += d.x;
      s1.y += d.y;
      s1.z += d.z;
      s1.w += d.w;
      s2.x += d.x * (v.x - mu.x) * is.x;
      s2.y += d.y * (v.y - mu.y) * is.y;
      s2.z += d.z * (v.z - mu.z) * is.z;
      s2.w += d.w * (v.w - mu.w) * is.w;
    }
  const float4 t1 = quad_sum(s1, q4, smem), t2 = quad_sum(s2, q4, smem);
  if ((int)threadIdx.x < q4) {
    float* p = partial + (long long)blockIdx.x * 2 * c + cq * 4;
    st4(p, t1);
    st4(p + c, t2);
  }
}

// one wave per output element (dbias[c], dweight[c]), lanes over the chunks, fixed butterfly: deterministic
__global__ void __launch_bounds__(64)
bn_bwd_merge_kernel(const float* __restrict__ partial, int nchunks, int c, float* __restrict__ dbias,
                    float* __restrict__ dweight) {
  const int lane = threadIdx.x, e = blockIdx.x;
  float s = 0.f;
  for (int b = lane; b < nchunks; b += 64) s += partial[(long long)b * 2 * c + e];
#pragma unroll
  for (int dlt = 32; dlt > 0; dlt >>= 1) s += __shfl_xor(s, dlt, 64);
  if (lane == 0) (e < c ? dbias : dweight)[e % c] = s;
}

__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                    const float* __restrict__ w, const float* __restrict__ mean, const float* __restrict__ invstd,
                    const float* __restrict__ dbias, const float* __restrict__ dweight, long long total4, long long m,
                    int c, int relu, float* __restrict__ dx, float* __restrict__ dres) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int ch = (int)((e * 4) % c);
  float4 d = ld4(dy + e * 4);
  const float4 v = ld4(x + e * 4);
  if (relu) {
    const float4 yy = ld4(y + e * 4);
    d.x = yy.x > 0.f ? d.x : 0.f;
    d.y = yy.y > 0.f ? d.y : 0.f;
    d.z = yy.z > 0.f ? d.z : 0.f;
    d.w = yy.w > 0.f ? d.w : 0.f;
  }
  if (dres) st4(dres + e * 4, d);
  const float4 mu = ld4(mean + ch), is = ld4(invstd + ch), ww = ld4(w + ch), a1 = ld4(dbias + ch), a2 = ld4(dweight + ch);
  const float inv_m = 1.0f / (float)m;
  float4 o;
  o.x = (d.x - a1.x * inv_m - (v.x - mu.x) * is.x * a2.x * inv_m) * is.x * ww.x;
  o.y = (d.y - a1.y * inv_m - (v.y - mu.y) * is.y * a2.y * inv_m) * is.y * ww.y;
  o.z = (d.z - a1.z * inv_m - (v.z - mu.z) * is.z * a2.z * inv_m) * is.z * ww.z;
  o.w = (d.w - a1.w * inv_m - (v.w - mu.w) * is.w * a2.w * inv_m) * is.w * ww.w;
  st4(dx + e * 4, o);
}

// ---- GroupNorm over channels-last BEV maps [B][rows = H*W][C] ------------------------------------------------
// (the input projection in front of the transformer, $CQ/voxel_detr.py:43-51: Conv2d 1x1 + GroupNorm(32, 256)).
// ATen's GroupNorm wants NCHW: on the channels-last map that costs a 72 MB transposing copy before it, one after it
// (the [B, HW, C] token layout the encoder reads) and the same two on the way back.  Here the statistics of a
// (sample, group) are the Chan merge of the per-chunk per-channel triples over the chunks and the group's channels.

// one wave per (group, sample): lanes over (chunk, channel-in-group) items
__global__ void __launch_bounds__(64)
gn_stats_merge_kernel(const float* __restrict__ partial, int nchunks, int c, int cpg, float eps,
                      float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x, g = blockIdx.x, b = blockIdx.y, groups = c / cpg;
  const float* base = partial + (long long)b * nchunks * 3 * c;
  float cnt = 0.f, mean = 0.f, M2 = 0.f;
  const int items = nchunks * cpg;
  for (int i0 = lane; i0 < items; i0 += 64 * 8) {   // eight items in flight per lane, folded in the same order
    float nbv[8], mbv[8], qbv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + 64 * u, items - 1);
      const float* p = base + (long long)(i / cpg) * 3 * c + g * cpg + i % cpg;
      nbv[u] = (i0 + 64 * u < items) ? p[0] : 0.f;
      mbv[u] = p[c];
      qbv[u] = p[2 * c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float nb = nbv[u], mb = mbv[u], qb = qbv[u];
      if (nb > 0.f) {
        const float tot_n = cnt + nb, d = mb - mean;
        mean += d * (nb / tot_n);
        M2 += qb + d * d * (cnt * nb / tot_n);
        cnt = tot_n;
      }
    }
  }
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const float n2 = __shfl_xor(cnt, dlt, 64), m2 = __shfl_xor(mean, dlt, 64), q2 = __shfl_xor(M2, dlt, 64);
    const bool low = !(lane & dlt);
    const float na = low ? cnt : n2, ma = low ? mean : m2, qa = low ? M2 : q2;
    const float nb = low ? n2 : cnt, mb = low ? m2 : mean, qb = low ? q2 : M2;
    const float tot_n = na + nb;
    if (tot_n > 0.f) {
      const float d = mb - ma;
      mean = ma + d * (nb / tot_n);
      M2 = qa + qb + d * d * (na * nb / tot_n);
    } else {
      mean = 0.f;
      M2 = 0.f;
    }
    cnt = tot_n;
  }
  if (lane == 0) {
    mean_out[b * groups + g] = mean;
    rstd_out[b * groups + g] = 1.0f / sqrtf((cnt > 0.f ? M2 / cnt : 0.f) + eps);
  }
}

// blockIdx.y = sample; e = float4 index inside the sample
__global__ void __launch_bounds__(256)
gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                const float* __restrict__ mean, const float* __restrict__ rstd, long long total4, int c, int cpg,
                float* __restrict__ y) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int ch = (int)((e * 4) % c), sg = blockIdx.y * (c / cpg) + ch / cpg;
  const long long o = ((long long)blockIdx.y * total4 + e) * 4;
  const float mu = mean[sg], rs = rstd[sg];
  const float4 v = ld4(x + o), ww = ld4(w + ch), bb = ld4(bias + ch);
  st4(y + o, make_float4((v.x - mu) * rs * ww.x + bb.x, (v.y - mu) * rs * ww.y + bb.y, (v.z - mu) * rs * ww.z + bb.z,
                         (v.w - mu) * rs * ww.w + bb.w));
}

// partial[sample][chunk][0][c] = sum dy, [1][c] = sum dy * xhat
__global__ void __launch_bounds__(256)
gn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                     const float* __restrict__ rstd, long long m, int c, int cpg, float* __restrict__ partial) {
  __shared__ float4 smem[256];
  const int q4 = c / 4, rstep = 256 / q4;
  const int cq = threadIdx.x % q4, r0 = threadIdx.x / q4;
  const bool lane_on = r0 < rstep;
  dy += (long long)blockIdx.y * m * c;
  x += (long long)blockIdx.y * m * c;
  partial += (long long)blockIdx.y * gridDim.x * 2 * c;
  const Chunk ck = chunk_of(m, gridDim.x, blockIdx.x);
  const int sg = blockIdx.y * (c / cpg) + (cq * 4) / cpg;
  const float mu = mean[sg], rs = rstd[sg];
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (lane_on)
    for (long long r = ck.row_lo + r0; r < ck.row_hi; r += rstep) {
      const long long o = r * c + cq * 4;
      const float4 d = ld4(dy + o), v = ld4(x + o);
      s1.x += d.x;
      s1.y += d.y;
      s1.z += d.z;
      s1.w += d.w;
      s2.x += d.x * (v.x - mu) * rs;
      s2.y += d.y * (v.y - mu) * rs;
      s2.z += d.z * (v.z - mu) * rs;
      s2.w += d.w * (v.w - mu) * rs;
    }
  const float4 t1 = quad_sum(s1, q4, smem), t2 = quad_sum(s2, q4, smem);
  if ((int)threadIdx.x < q4) {
    float* p = partial + (long long)blockIdx.x * 2 * c + cq * 4;
    st4(p, t1);
    st4(p + c, t2);
  }
}

// one wave per (element of [2][c], sample): sums[sample][2][c], lanes over the chunks
__global__ void __launch_bounds__(64)
gn_bwd_merge_kernel(const float* __restrict__ partial, int nchunks, int c, float* __restrict__ sums) {
  const int lane = threadIdx.x, e = blockIdx.x, b = blockIdx.y;
  const float* base = partial + (long long)b * nchunks * 2 * c;
  float s = 0.f;
  for (int k = lane; k < nchunks; k += 64) s += base[(long long)k * 2 * c + e];
#pragma unroll
  for (int dlt = 32; dlt > 0; dlt >>= 1) s += __shfl_xor(s, dlt, 64);
  if (lane == 0) sums[(long long)b * 2 * c + e] = s;
}

// dbias / dweight over the samples; per (sample, group): A = sum_c w_c S1, B = sum_c w_c S2  (ab[sample][group][2])
__global__ void __launch_bounds__(256)
gn_bwd_finish_kernel(const float* __restrict__ sums, const float* __restrict__ w, int batch, int c, int cpg,
                     float* __restrict__ dbias, float* __restrict__ dweight, float* __restrict__ ab) {
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < batch; ++b) {
      s1 += sums[(long long)b * 2 * c + ch];
      s2 += sums[(long long)b * 2 * c + c + ch];
    }
    dbias[ch] = s1;
    dweight[ch] = s2;
  }
  const int groups = c / cpg;
  for (int i = threadIdx.x; i < batch * groups; i += 256) {
    const int b = i / groups, g = i % groups;
    float a = 0.f, bb = 0.f;
    for (int k = 0; k < cpg; ++k) {
      const int ch = g * cpg + k;
      a += w[ch] * sums[(long long)b * 2 * c + ch];
      bb += w[ch] * sums[(long long)b * 2 * c + c + ch];
    }
    ab[2 * i] = a;
    ab[2 * i + 1] = bb;
  }
}

__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ ab,
                    long long total4, long long m, int c, int cpg, float* __restrict__ dx) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int ch = (int)((e * 4) % c), sg = blockIdx.y * (c / cpg) + ch / cpg;
  const long long o = ((long long)blockIdx.y * total4 + e) * 4;
  const float mu = mean[sg], rs = rstd[sg];
  const float inv_n = 1.0f / ((float)m * (float)cpg);
  const float a = ab[2 * sg] * inv_n, b = ab[2 * sg + 1] * inv_n;
  const float4 d = ld4(dy + o), v = ld4(x + o), ww = ld4(w + ch);
  st4(dx + o, make_float4((ww.x * d.x - a - (v.x - mu) * rs * b) * rs, (ww.y * d.y - a - (v.y - mu) * rs * b) * rs,
                          (ww.z * d.z - a - (v.z - mu) * rs * b) * rs, (ww.w * d.w - a - (v.w - mu) * rs * b) * rs));
}

int nblocks_for(long long m) { return (int)std::max<long long>(1, std::min<long long>(kBnBlocks, ceil_div(m, 32))); }

}  // namespace
}  // namespace efg

using namespace efg;

// workspace: partials [kBnBlocks][3][c] floats
extern "C" size_t efg_bn_workspace_bytes(int c) {
  if (c < 1) return 0;
  return align_up(sizeof(float) * 3 * (size_t)c * kBnBlocks, 256) + 256;
}

static int bn_check(int64_t m, int c) {
  EFG_CHECK_ARG(m >= 0 && c >= 4 && c % 4 == 0 && c <= 1024, "batch_norm: need c %% 4 == 0 and c <= 1024 (got %d)", c);
  return EFG_OK;
}

extern "C" int efg_bn_forward_f32(const float* x, const float* residual, const float* weight, const float* bias,
                                  float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                  float eps, int64_t m, int c, int relu, float* y, float* mean, float* invstd, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (int rc = bn_check(m, c)) return rc;
  EFG_CHECK_ARG(m >= 1, "batch_norm: training statistics need at least one row");
  EFG_CHECK_ARG(x && weight && bias && y && mean && invstd && ws, "batch_norm: null pointer");
  EFG_CHECK_ARG(ws_bytes >= efg_bn_workspace_bytes(c), "batch_norm: workspace too small");
  EFG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "batch_norm: running_mean / running_var go together");
  hipStream_t st = (hipStream_t)stream;
  const int nb = nblocks_for(m);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nb), dim3(256), 0, st, x, (long long)m, c, static_cast<float*>(ws));
  hipLaunchKernelGGL(bn_stats_merge_kernel, dim3(c), dim3(64), 0, st, static_cast<const float*>(ws), nb, c, eps, momentum,
                     mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
  EFG_LAUNCH_CHECK();
  const long long total4 = (long long)m * c / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, x, residual, weight, bias,
                     mean, invstd, total4, c, relu, y);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_bn_backward_f32(const float* dy, const float* x, const float* y, const float* weight, const float* mean,
                                   const float* invstd, int64_t m, int c, int relu, float* dx, float* dresidual,
                                   float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = bn_check(m, c)) return rc;
  EFG_CHECK_ARG(m >= 1, "batch_norm backward: empty input");
  EFG_CHECK_ARG(dy && x && weight && mean && invstd && dx && dweight && dbias && ws && (!relu || y),
                "batch_norm backward: null pointer");
  EFG_CHECK_ARG(ws_bytes >= efg_bn_workspace_bytes(c), "batch_norm backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = nblocks_for(m);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, dy, x, y, mean, invstd, (long long)m, c, relu,
                     static_cast<float*>(ws));
  hipLaunchKernelGGL(bn_bwd_merge_kernel, dim3(2 * c), dim3(64), 0, st, static_cast<const float*>(ws), nb, c, dbias, dweight);
  EFG_LAUNCH_CHECK();
  const long long total4 = (long long)m * c / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, dy, x, y, weight, mean,
                     invstd, dbias, dweight, total4, (long long)m, c, relu, dx, dresidual);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

// ---- GroupNorm (channels-last) --------------------------------------------------------------------------------
// workspace: partials [batch][kBnBlocks][3][c] + sums [batch][2][c] + ab [batch][groups][2] floats
extern "C" size_t efg_gn_workspace_bytes(int batch, int c) {
  if (batch < 1 || c < 1) return 0;
  return align_up(sizeof(float) * ((size_t)batch * kBnBlocks * 3 * c + (size_t)batch * 4 * c), 256) + 256;
}

static int gn_check(int batch, int64_t rows, int c, int groups) {
  EFG_CHECK_ARG(batch >= 1 && batch <= 65535 && rows >= 1, "group_norm: need batch in [1, 65535] and rows >= 1");
  EFG_CHECK_ARG(c >= 4 && c % 4 == 0 && c <= 1024, "group_norm: need c %% 4 == 0 and c <= 1024 (got %d)", c);
  EFG_CHECK_ARG(groups >= 1 && c % groups == 0 && (c / groups) % 4 == 0,
                "group_norm: channels per group must be a multiple of 4 (c = %d, groups = %d)", c, groups);
  return EFG_OK;
}

extern "C" int efg_gn_forward_f32(const float* x, const float* weight, const float* bias, float eps, int batch,
                                  int64_t rows, int c, int groups, float* y, float* mean, float* rstd, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (int rc = gn_check(batch, rows, c, groups)) return rc;
  EFG_CHECK_ARG(x && weight && bias && y && mean && rstd && ws, "group_norm: null pointer");
  EFG_CHECK_ARG(ws_bytes >= efg_gn_workspace_bytes(batch, c), "group_norm: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = nblocks_for(rows), cpg = c / groups;
  float* partial = static_cast<float*>(ws);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nb, batch), dim3(256), 0, st, x, (long long)rows, c, partial);
  hipLaunchKernelGGL(gn_stats_merge_kernel, dim3(groups, batch), dim3(64), 0, st, partial, nb, c, cpg, eps, mean, rstd);
  EFG_LAUNCH_CHECK();
  const long long total4 = (long long)rows * c / 4;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)ceil_div(total4, 256), batch), dim3(256), 0, st, x, weight, bias,
                     mean, rstd, total4, c, cpg, y);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_gn_backward_f32(const float* dy, const float* x, const float* weight, const float* mean,
                                   const float* rstd, int batch, int64_t rows, int c, int groups, float* dx,
                                   float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = gn_check(batch, rows, c, groups)) return rc;
  EFG_CHECK_ARG(dy && x && weight && mean && rstd && dx && dweight && dbias && ws, "group_norm backward: null pointer");
  EFG_CHECK_ARG(ws_bytes >= efg_gn_workspace_bytes(batch, c), "group_norm backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = nblocks_for(rows), cpg = c / groups;
  float* partial = static_cast<float*>(ws);
  float* sums = partial + (size_t)batch * kBnBlocks * 3 * c;
  float* ab = sums + (size_t)batch * 2 * c;
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(nb, batch), dim3(256), 0, st, dy, x, mean, rstd, (long long)rows, c, cpg,
                     partial);
  hipLaunchKernelGGL(gn_bwd_merge_kernel, dim3(2 * c, batch), dim3(64), 0, st, partial, nb, c, sums);
  hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3(1), dim3(256), 0, st, sums, weight, batch, c, cpg, dbias, dweight, ab);
  EFG_LAUNCH_CHECK();
  const long long total4 = (long long)rows * c / 4;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)ceil_div(total4, 256), batch), dim3(256), 0, st, dy, x, weight,
                     mean, rstd, ab, total4, (long long)rows, c, cpg, dx);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
