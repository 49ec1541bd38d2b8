// Fused residual-add + LayerNorm for gfx950 (forward and backward).
//
// The transformer layers of the path are post-norm: x = norm(x + sublayer(x)) ($CQ/transformer.py:231-243,
// 296-317).  On the 70 688-token encoder sequence (72 MB per tensor) PyTorch runs that as an add kernel, a
// LayerNorm kernel and, backward, three more passes (grad-input, two-stage gamma/beta reduction).  Here:
//   forward : one pass  -- z = x + r (stored once: it is the saved tensor), y = LN(z), mean / rstd per row
//   backward: one pass  -- dz (which is the gradient of both x and r) + per-workgroup partial dgamma / dbeta,
//             and a tiny second kernel that sums the partials in a fixed order (deterministic).
// One wave per row (C / 4 <= 64 float4 lanes per pass, C <= 1024), reductions with DPP-class shuffles, no LDS in
// the forward.  HBM-bound: 3 (fwd) and 3 (bwd) tensor passes.
#include "common.h"

#include <algorithm>

namespace efg {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// VPL: float4 per lane (C <= 256 * VPL); RPW: rows per wave handled together -- their loads are issued back to
// back, which is what keeps enough bytes in flight for a pass that is pure HBM streaming.
template <int VPL, int RPW>
__global__ void __launch_bounds__(256)
add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ gamma,
                  const float* __restrict__ beta, float eps, long long rows, int C, float* __restrict__ z_out,
                  float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  float4 z[RPW][VPL];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const long long row = min(row0 + k, rows - 1);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = min((lane + 64 * i) * 4, C - 4);
      z[k][i] = ld4(x + row * C + c);
    }
  }
  if (r) {
    float4 rv[RPW][VPL];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const long long row = min(row0 + k, rows - 1);
#pragma unroll
      for (int i = 0; i < VPL; ++i) rv[k][i] = ld4(r + row * C + min((lane + 64 * i) * 4, C - 4));
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        z[k][i].x += rv[k][i].x;
        z[k][i].y += rv[k][i].y;
        z[k][i].z += rv[k][i].z;
        z[k][i].w += rv[k][i].w;
      }
  }
  float4 g[VPL], bt[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = min((lane + 64 * i) * 4, C - 4);
    g[i] = ld4(gamma + c);
    bt[i] = ld4(beta + c);
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const long long row = row0 + k;
    if (row >= rows) break;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if ((lane + 64 * i) * 4 < C) s += (z[k][i].x + z[k][i].y) + (z[k][i].z + z[k][i].w);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if ((lane + 64 * i) * 4 < C) {
        const float a = z[k][i].x - mean, b = z[k][i].y - mean, cc = z[k][i].z - mean, d = z[k][i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < C) {
        if (r) st4(z_out + row * C + c, z[k][i]);
        float4 o;
        o.x = (z[k][i].x - mean) * rstd * g[i].x + bt[i].x;
        o.y = (z[k][i].y - mean) * rstd * g[i].y + bt[i].y;
        o.z = (z[k][i].z - mean) * rstd * g[i].z + bt[i].z;
        o.w = (z[k][i].w - mean) * rstd * g[i].w + bt[i].w;
        st4(y + row * C + c, o);
      }
    }
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// grid-stride over groups of RPW rows per wave; partial[blockIdx][0][C] = dgamma, [1][C] = dbeta of the block
template <int VPL, int RPW>
__global__ void __launch_bounds__(256)
add_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean,
                  const float* __restrict__ rstd, const float* __restrict__ gamma, long long rows, int C,
                  float* __restrict__ dz, float* __restrict__ partial) {
  __shared__ float red[4][2][256 * VPL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float4 g[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    g[i] = c < C ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invC = 1.0f / (float)C;
  for (long long row0 = ((long long)blockIdx.x * 4 + wv) * RPW; row0 < rows; row0 += (long long)gridDim.x * 4 * RPW) {
    float4 zv[RPW][VPL], d[RPW][VPL];
    float mu[RPW], rs[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const long long row = min(row0 + k, rows - 1);
      mu[k] = mean[row];
      rs[k] = rstd[row];
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = min((lane + 64 * i) * 4, C - 4);
        zv[k][i] = ld4(z + row * C + c);
        d[k][i] = ld4(dy + row * C + c);
      }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const long long row = row0 + k;
      if (row >= rows) break;
      float4 xh[VPL], gy[VPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const bool on = (lane + 64 * i) * 4 < C;
        const float4 dd = on ? d[k][i] : make_float4(0.f, 0.f, 0.f, 0.f);
        xh[i] = make_float4((zv[k][i].x - mu[k]) * rs[k], (zv[k][i].y - mu[k]) * rs[k], (zv[k][i].z - mu[k]) * rs[k],
                            (zv[k][i].w - mu[k]) * rs[k]);
        gy[i] = make_float4(dd.x * g[i].x, dd.y * g[i].y, dd.z * g[i].z, dd.w * g[i].w);
        s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
        s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
        dg[i].x += dd.x * xh[i].x;
        dg[i].y += dd.y * xh[i].y;
        dg[i].z += dd.z * xh[i].z;
        dg[i].w += dd.w * xh[i].w;
        db[i].x += dd.x;
        db[i].y += dd.y;
        db[i].z += dd.z;
        db[i].w += dd.w;
      }
      const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < C) {
          float4 o;
          o.x = rs[k] * (gy[i].x - m1 - xh[i].x * m2);
          o.y = rs[k] * (gy[i].y - m1 - xh[i].y * m2);
          o.z = rs[k] * (gy[i].z - m1 - xh[i].z * m2);
          o.w = rs[k] * (gy[i].w - m1 - xh[i].w * m2);
          st4(dz + row * C + c, o);
        }
      }
    }
  }
  // block partials: the 4 waves meet in LDS, then one row of `partial` per block
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    st4(&red[wv][0][c], dg[i]);
    st4(&red[wv][1][c], db[i]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += 256) {
    const int which = e / C, c = e % C;
    partial[((long long)blockIdx.x * 2 + which) * C + c] =
        (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// one wave per output element: lanes stride over the block partials (fixed order -> deterministic)
__global__ void __launch_bounds__(256)
ln_param_grad_kernel(const float* __restrict__ partial, int nblocks, int C, float* __restrict__ dgamma,
                     float* __restrict__ dbeta) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= 2 * C) return;
  const int which = e / C, c = e % C;
  float s = 0.f;
  for (int b = lane; b < nblocks; b += 64) s += partial[((long long)b * 2 + which) * C + c];
  s = wave_sum(s);
  if (lane == 0) (which ? dbeta : dgamma)[c] = s;
}

int vpl_of(int C) { return C <= 256 ? 1 : C <= 512 ? 2 : 4; }
constexpr int kRPW = 4;  // rows per wave and pass
int bwd_blocks(long long rows) { return (int)std::min<long long>(ceil_div(rows, 4 * kRPW), 2048); }

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_add_layernorm_forward_f32(const float* x, const float* residual, const float* gamma,
                                             const float* beta, float eps, int64_t rows, int c, float* z_out,
                                             float* y, float* mean, float* rstd, void* stream) {
  EFG_CHECK_ARG(rows >= 0 && c >= 4 && c % 4 == 0 && c <= 1024, "add_layernorm: need c %% 4 == 0 and c <= 1024, got %d", c);
  if (rows == 0) return EFG_OK;
  EFG_CHECK_ARG(x && gamma && beta && y && mean && rstd, "add_layernorm: null pointer");
  EFG_CHECK_ARG(!residual || z_out, "add_layernorm: z_out is required when a residual is given");
  const dim3 grid((unsigned)ceil_div(rows, 4 * kRPW)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (vpl_of(c)) {
    case 1: hipLaunchKernelGGL((add_ln_fwd_kernel<1, kRPW>), grid, block, 0, st, x, residual, gamma, beta, eps, rows, c, z_out, y, mean, rstd); break;
    case 2: hipLaunchKernelGGL((add_ln_fwd_kernel<2, kRPW>), grid, block, 0, st, x, residual, gamma, beta, eps, rows, c, z_out, y, mean, rstd); break;
    default: hipLaunchKernelGGL((add_ln_fwd_kernel<4, kRPW>), grid, block, 0, st, x, residual, gamma, beta, eps, rows, c, z_out, y, mean, rstd); break;
  }
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" size_t efg_add_layernorm_backward_workspace_bytes(int64_t rows, int c) {
  if (rows < 0 || c < 1) return 0;
  return align_up(sizeof(float) * 2 * (size_t)c * (size_t)bwd_blocks(rows > 0 ? rows : 1), 256);
}

extern "C" int efg_add_layernorm_backward_f32(const float* dy, const float* z, const float* mean, const float* rstd,
                                              const float* gamma, int64_t rows, int c, float* dz, float* dgamma,
                                              float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  EFG_CHECK_ARG(rows >= 0 && c >= 4 && c % 4 == 0 && c <= 1024, "add_layernorm: need c %% 4 == 0 and c <= 1024, got %d", c);
  EFG_CHECK_ARG(dgamma && dbeta, "add_layernorm backward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    EFG_HIP_TRY(hipMemsetAsync(dgamma, 0, sizeof(float) * c, st));
    EFG_HIP_TRY(hipMemsetAsync(dbeta, 0, sizeof(float) * c, st));
    return EFG_OK;
  }
  EFG_CHECK_ARG(dy && z && mean && rstd && gamma && dz && ws, "add_layernorm backward: null pointer");
  EFG_CHECK_ARG(ws_bytes >= efg_add_layernorm_backward_workspace_bytes(rows, c), "add_layernorm backward: workspace too small");
  const int nb = bwd_blocks(rows);
  float* partial = static_cast<float*>(ws);
  switch (vpl_of(c)) {
    case 1: hipLaunchKernelGGL((add_ln_bwd_kernel<1, kRPW>), dim3(nb), dim3(256), 0, st, dy, z, mean, rstd, gamma, rows, c, dz, partial); break;
    case 2: hipLaunchKernelGGL((add_ln_bwd_kernel<2, kRPW>), dim3(nb), dim3(256), 0, st, dy, z, mean, rstd, gamma, rows, c, dz, partial); break;
    default: hipLaunchKernelGGL((add_ln_bwd_kernel<4, kRPW>), dim3(nb), dim3(256), 0, st, dy, z, mean, rstd, gamma, rows, c, dz, partial); break;
  }
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3((unsigned)ceil_div(2 * c, 4)), dim3(256), 0, st, partial, nb, c, dgamma, dbeta);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
