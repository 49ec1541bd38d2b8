// Sparse-convolution GEOMETRY for gfx950: active-site index, output-site discovery, neighbour
// ("rulebook") tables, dense <-> sparse.
//
// Replaces the indice-pair generation and .dense() of third-party spconv as used by
// efg/modeling/backbones/sparse_net.py:79-98,120-165,273-309,485-545 (contract SURVEY.md B.6).
// Design (not spconv's hash-table pair lists): every level of the backbone keeps a succinct
// RANK INDEX (rank_index.h) over its linearised (b,z,y,x) grid.  Output sites of a strided conv
// are discovered by OR-ing bits and ranked by a popcount scan, which yields them in canonical
// ascending order with no sort; neighbour lookup is one 8-byte load per probe.  The rulebook is
// OUTPUT-STATIONARY: nbr[k][o] = input row feeding output o through kernel offset k (or -1), so
// the conv kernels need no atomics and are deterministic; rnbr is its transpose for dgrad.
#include "rank_index.h"

namespace efg {
namespace {

struct Grid3 {
  int b, d, h, w;
};

struct ConvGeom {
  int k[3], s[3], p[3];
};

__device__ __forceinline__ unsigned long long cell_of(const Grid3& g, int b, int z, int y, int x) {
  return (((unsigned long long)b * g.d + z) * g.h + y) * g.w + x;
}

__global__ void __launch_bounds__(256) idx_mark_kernel(const int* __restrict__ ind, long long m, Grid3 g,
                                                        uint2* __restrict__ idx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(ind)[i];
    if ((unsigned)c.x < (unsigned)g.b && (unsigned)c.y < (unsigned)g.d && (unsigned)c.z < (unsigned)g.h &&
        (unsigned)c.w < (unsigned)g.w)
      rank_set(idx, cell_of(g, c.x, c.y, c.z, c.w));
  }
}

__global__ void __launch_bounds__(256) idx_perm_kernel(const int* __restrict__ ind, long long m, Grid3 g,
                                                        const uint2* __restrict__ idx, int* __restrict__ perm) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(ind)[i];
    if ((unsigned)c.x < (unsigned)g.b && (unsigned)c.y < (unsigned)g.d && (unsigned)c.z < (unsigned)g.h &&
        (unsigned)c.w < (unsigned)g.w) {
      const int r = rank_lookup(idx, cell_of(g, c.x, c.y, c.z, c.w));
      if (r >= 0) perm[r] = (int)i;
    }
  }
}

// every input site marks the output sites it feeds: o = (i + p - k) / s when divisible, in range
__global__ void __launch_bounds__(256)
idx_downsample_mark_kernel(const int* __restrict__ ind, long long m, Grid3 gin, Grid3 gout, ConvGeom cg,
                           uint2* __restrict__ oidx, int* __restrict__ count) {
  // `fresh` counts the output sites this thread switched on: their sum is the number of output sites, available
  // as soon as this kernel is done (the host sizes the next level with it while the ranking kernels still run)
  int fresh = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(ind)[i];
    if (!((unsigned)c.x < (unsigned)gin.b && (unsigned)c.y < (unsigned)gin.d && (unsigned)c.z < (unsigned)gin.h &&
          (unsigned)c.w < (unsigned)gin.w))
      continue;
    for (int kz = 0; kz < cg.k[0]; ++kz) {
      const int tz = c.y + cg.p[0] - kz;
      if (tz < 0 || tz % cg.s[0]) continue;
      const int oz = tz / cg.s[0];
      if (oz >= gout.d) continue;
      for (int ky = 0; ky < cg.k[1]; ++ky) {
        const int ty = c.z + cg.p[1] - ky;
        if (ty < 0 || ty % cg.s[1]) continue;
        const int oy = ty / cg.s[1];
        if (oy >= gout.h) continue;
        for (int kx = 0; kx < cg.k[2]; ++kx) {
          const int tx = c.w + cg.p[2] - kx;
          if (tx < 0 || tx % cg.s[2]) continue;
          const int ox = tx / cg.s[2];
          if (ox >= gout.w) continue;
          fresh += rank_set_new(oidx, cell_of(gout, c.x, oz, oy, ox)) ? 1 : 0;
        }
      }
    }
  }
  fresh = wave_reduce_sum(fresh);
  if (lane_id() == 0 && fresh) atomicAdd(count, fresh);
}

// one thread per index word: write the (b,z,y,x) rows of its set bits at their ranks
__global__ void __launch_bounds__(256) idx_emit_kernel(const uint2* __restrict__ idx, long long words, Grid3 g,
                                                        int* __restrict__ out) {
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < words;
       w += (long long)gridDim.x * blockDim.x) {
    const uint2 u = idx[w];
    unsigned bits = u.x;
    unsigned r = u.y;
    while (bits) {
      const int bpos = __ffs(bits) - 1;
      bits &= bits - 1;
      unsigned long long cell = (unsigned long long)w * 32 + bpos;
      int4 c;
      c.w = (int)(cell % g.w);
      cell /= g.w;
      c.z = (int)(cell % g.h);
      cell /= g.h;
      c.y = (int)(cell % g.d);
      c.x = (int)(cell / g.d);
      reinterpret_cast<int4*>(out)[r++] = c;
    }
  }
}

// nbr[k][o]: one thread per output site walks the kernel window
__global__ void __launch_bounds__(256)
nbr_kernel(const uint2* __restrict__ iidx, const int* __restrict__ iperm, Grid3 gin, const int* __restrict__ oind,
           long long m_out, ConvGeom cg, int* __restrict__ nbr) {
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < m_out;
       o += (long long)gridDim.x * blockDim.x) {
    const int4 c = reinterpret_cast<const int4*>(oind)[o];
    int k = 0;
    for (int kz = 0; kz < cg.k[0]; ++kz) {
      const int iz = c.y * cg.s[0] - cg.p[0] + kz;
      for (int ky = 0; ky < cg.k[1]; ++ky) {
        const int iy = c.z * cg.s[1] - cg.p[1] + ky;
        const bool row_ok = (unsigned)iz < (unsigned)gin.d && (unsigned)iy < (unsigned)gin.h;
        for (int kx = 0; kx < cg.k[2]; ++kx, ++k) {
          const int ix = c.w * cg.s[2] - cg.p[2] + kx;
          int r = -1;
          if (row_ok && (unsigned)ix < (unsigned)gin.w) {
            r = rank_lookup(iidx, cell_of(gin, c.x, iz, iy, ix));
            if (r >= 0 && iperm) r = iperm[r];
          }
          nbr[(long long)k * m_out + o] = r;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) rnbr_kernel(const int* __restrict__ nbr, long long m_out, int kvol,
                                                    long long m_in, int* __restrict__ rnbr) {
  const long long total = m_out * kvol;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int i = nbr[e];
    if (i >= 0) {
      const long long k = e / m_out;
      rnbr[k * m_in + i] = (int)(e - k * m_out);
    }
  }
}

// dense[b][c][sp] for a tile of 64 consecutive spatial cells x 64 channels, transposed through LDS:
// feature rows are read as contiguous 256-byte runs, dense rows are written as 256-byte runs.
__global__ void __launch_bounds__(256)
to_dense_kernel(const float* __restrict__ feat, int c, const uint2* __restrict__ idx, const int* __restrict__ perm,
                long long dhw, float* __restrict__ dense) {
  __shared__ float tile[64][65];
  __shared__ int rows[64];
  __shared__ int any_row;
  const int b = blockIdx.y;
  const long long sp0 = (long long)blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) any_row = 0;
  __syncthreads();
  if (wv == 0) {
    int r = -1;
    if (sp0 + lane < dhw) {
      r = rank_lookup(idx, (unsigned long long)b * dhw + sp0 + lane);
      if (r >= 0 && perm) r = perm[r];
    }
    rows[lane] = r;
    if (r >= 0) any_row = 1;
  }
  __syncthreads();
  const bool any = any_row != 0;
  for (int c0 = 0; c0 < c; c0 += 64) {
    if (any) {
      for (int j = wv; j < 64; j += 4) {
        const int r = rows[j];
        tile[j][lane] = (r >= 0 && c0 + lane < c) ? feat[(long long)r * c + c0 + lane] : 0.0f;
      }
      __syncthreads();
    }
    for (int cc = wv; cc < 64 && c0 + cc < c; cc += 4)
      if (sp0 + lane < dhw) dense[((long long)b * c + c0 + cc) * dhw + sp0 + lane] = any ? tile[lane][cc] : 0.0f;
    if (any) __syncthreads();
  }
}

// grad_feat[row][c] = grad_dense[b][c][sp(row)], 64 rows x 64 channels per tile through LDS
__global__ void __launch_bounds__(256)
from_dense_kernel(const float* __restrict__ gd, int c, const int* __restrict__ ind, long long m, Grid3 g,
                  float* __restrict__ gf) {
  __shared__ float tile[64][65];
  __shared__ long long base[64];
  const long long r0 = (long long)blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long dhw = (long long)g.d * g.h * g.w;
  if (wv == 0) {
    long long bs = -1;
    if (r0 + lane < m) {
      const int4 q = reinterpret_cast<const int4*>(ind)[r0 + lane];
      bs = (long long)q.x * c * dhw + ((long long)q.y * g.h + q.z) * g.w + q.w;
    }
    base[lane] = bs;
  }
  __syncthreads();
  for (int c0 = 0; c0 < c; c0 += 64) {
    const long long bs = base[lane];
    for (int cc = wv; cc < 64; cc += 4)
      tile[lane][cc] = (bs >= 0 && c0 + cc < c) ? gd[bs + (long long)(c0 + cc) * dhw] : 0.0f;
    __syncthreads();
    for (int j = wv; j < 64; j += 4)
      if (r0 + j < m && c0 + lane < c) gf[(r0 + j) * c + c0 + lane] = tile[j][lane];
    __syncthreads();
  }
}


// BEV flatten fused with densify: out[b][y][x][c*D + d] = feat[row(b,d,y,x)][c] (0 if inactive), i.e.
// SparseConvTensor.dense() -> view(N, C*D, H, W) of the reference (sparse_net.py:304-306) written
// directly in channels-last (NHWC) order, the layout the transformer consumes ([B, H*W, C] tokens).
// One workgroup = 8 consecutive BEV pixels; writes are fully coalesced runs of C*D floats.
constexpr int kBevPix = 8;
constexpr int kBevMaxD = 16;
__global__ void __launch_bounds__(256)
to_bev_kernel(const float* __restrict__ feat, int c, const uint2* __restrict__ idx, const int* __restrict__ perm,
              Grid3 g, float* __restrict__ out) {
  __shared__ int rows[kBevPix][kBevMaxD];
  const long long hw = (long long)g.h * g.w;
  const long long npix = (long long)g.b * hw;
  const long long p0 = (long long)blockIdx.x * kBevPix;
  const int cd = c * g.d;
  if (threadIdx.x < kBevPix * g.d) {
    const int pix = threadIdx.x / g.d, d = threadIdx.x % g.d;
    int r = -1;
    if (p0 + pix < npix) {
      const long long b = (p0 + pix) / hw, yx = (p0 + pix) % hw;
      r = rank_lookup(idx, ((unsigned long long)b * g.d + d) * hw + yx);
      if (r >= 0 && perm) r = perm[r];
    }
    rows[pix][d] = r;
  }
  __syncthreads();
  const int total = kBevPix * cd;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int pix = e / cd, ch = e - pix * cd;
    if (p0 + pix >= npix) break;
    const int cc = ch / g.d, d = ch - cc * g.d;
    const int r = rows[pix][d];
    out[(p0 + pix) * cd + ch] = (r >= 0) ? feat[(long long)r * c + cc] : 0.0f;
  }
}

__global__ void __launch_bounds__(256)
from_bev_kernel(const float* __restrict__ gout, int c, const int* __restrict__ ind, long long m, Grid3 g,
                float* __restrict__ gf) {
  const long long total = m * c;
  const int cd = c * g.d;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / c;
    const int cc = (int)(e - r * c);
    const int4 q = reinterpret_cast<const int4*>(ind)[r];
    gf[e] = gout[(((long long)q.x * g.h + q.z) * g.w + q.w) * cd + cc * g.d + q.y];
  }
}

int make_grid(int batch, const int* shape, Grid3* g, unsigned long long* cells) {
  EFG_CHECK_ARG(batch >= 1 && shape[0] >= 1 && shape[1] >= 1 && shape[2] >= 1, "spconv: bad grid shape");
  *g = Grid3{batch, shape[0], shape[1], shape[2]};
  const unsigned long long c = (unsigned long long)batch * shape[0] * shape[1] * shape[2];
  EFG_CHECK_ARG(c < 0xffffffffull, "spconv: batch x grid must be < 2^32-1 cells");
  *cells = c;
  return EFG_OK;
}

int make_conv(const int* k, const int* s, const int* p, ConvGeom* cg) {
  for (int a = 0; a < 3; ++a) {
    EFG_CHECK_ARG(k[a] >= 1 && s[a] >= 1 && p[a] >= 0, "spconv: bad kernel/stride/padding");
    cg->k[a] = k[a];
    cg->s[a] = s[a];
    cg->p[a] = p[a];
  }
  EFG_CHECK_ARG(k[0] * k[1] * k[2] <= 125, "spconv: kernel volume > 125 not supported");
  return EFG_OK;
}

inline int grid_for(long long work) { return (int)std::min<long long>(std::max<long long>(ceil_div(work, 256), 1), 8192); }

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" size_t efg_spconv_index_bytes(int batch, const int* shape) {
  Grid3 g;
  unsigned long long cells;
  if (make_grid(batch, shape, &g, &cells) != EFG_OK) return 0;
  return (size_t)rank_words(cells) * 8;
}

extern "C" size_t efg_spconv_index_workspace_bytes(int batch, const int* shape) {
  Grid3 g;
  unsigned long long cells;
  if (make_grid(batch, shape, &g, &cells) != EFG_OK) return 0;
  return align_up((size_t)rank_tiles(rank_words(cells)) * 4, 256) + 256;
}

extern "C" int efg_spconv_index_from_indices(const int32_t* indices, int64_t m, int batch, const int* shape,
                                             void* index, int32_t* perm, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Grid3 g;
  unsigned long long cells;
  if (int rc = make_grid(batch, shape, &g, &cells)) return rc;
  EFG_CHECK_ARG(m >= 0 && m < (1ll << 31), "spconv: bad row count");
  const long long words = rank_words(cells);
  Workspace w(ws, ws_bytes);
  int* tile_sums = w.take<int>(rank_tiles(words));
  if (!w.ok) {
    set_error("spconv index workspace too small");
    return EFG_E_WORKSPACE;
  }
  uint2* idx = static_cast<uint2*>(index);
  EFG_HIP_TRY(hipMemsetAsync(idx, 0, (size_t)words * 8, stream));
  if (m > 0) {
    hipLaunchKernelGGL(idx_mark_kernel, dim3(grid_for(m)), dim3(256), 0, stream, indices, (long long)m, g, idx);
    EFG_LAUNCH_CHECK();
  }
  if (int rc = rank_build_prefix(idx, words, tile_sums, nullptr, stream)) return rc;
  if (m > 0 && perm) {
    hipLaunchKernelGGL(idx_perm_kernel, dim3(grid_for(m)), dim3(256), 0, stream, indices, (long long)m, g, idx, perm);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

extern "C" int efg_spconv_index_downsample(const int32_t* in_indices, int64_t m_in, int batch, const int* in_shape,
                                           const int* ksize, const int* stride, const int* pad, void* out_index,
                                           int* out_shape, int32_t* m_out_dev, void* ws, size_t ws_bytes,
                                           void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom cg;
  if (int rc = make_conv(ksize, stride, pad, &cg)) return rc;
  for (int a = 0; a < 3; ++a) {
    out_shape[a] = (in_shape[a] + 2 * pad[a] - ksize[a]) / stride[a] + 1;
    EFG_CHECK_ARG(in_shape[a] + 2 * pad[a] >= ksize[a], "spconv: kernel larger than padded input on axis %d", a);
  }
  Grid3 gin, gout;
  unsigned long long cin, cout;
  if (int rc = make_grid(batch, in_shape, &gin, &cin)) return rc;
  if (int rc = make_grid(batch, out_shape, &gout, &cout)) return rc;
  const long long words = rank_words(cout);
  Workspace w(ws, ws_bytes);
  int* tile_sums = w.take<int>(rank_tiles(words));
  if (!w.ok) {
    set_error("spconv index workspace too small");
    return EFG_E_WORKSPACE;
  }
  (void)tile_sums;
  uint2* oidx = static_cast<uint2*>(out_index);
  EFG_HIP_TRY(hipMemsetAsync(oidx, 0, (size_t)words * 8, stream));
  EFG_HIP_TRY(hipMemsetAsync(m_out_dev, 0, sizeof(int32_t), stream));
  if (m_in > 0) {
    hipLaunchKernelGGL(idx_downsample_mark_kernel, dim3(grid_for(m_in)), dim3(256), 0, stream, in_indices,
                       (long long)m_in, gin, gout, cg, oidx, m_out_dev);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

extern "C" int efg_spconv_index_rank(void* index, int batch, const int* shape, int32_t* m_dev, void* ws, size_t ws_bytes,
                                     void* stream_) {
  Grid3 g;
  unsigned long long cells;
  if (int rc = make_grid(batch, shape, &g, &cells)) return rc;
  const long long words = rank_words(cells);
  Workspace w(ws, ws_bytes);
  int* tile_sums = w.take<int>(rank_tiles(words));
  int* total = w.take<int>(1);
  if (!w.ok) {
    set_error("spconv index workspace too small");
    return EFG_E_WORKSPACE;
  }
  return rank_build_prefix(static_cast<uint2*>(index), words, tile_sums, m_dev ? m_dev : total, (hipStream_t)stream_);
}

extern "C" int efg_spconv_index_emit(const void* index, int batch, const int* shape, int32_t* out_indices,
                                     void* stream_) {
  Grid3 g;
  unsigned long long cells;
  if (int rc = make_grid(batch, shape, &g, &cells)) return rc;
  const long long words = rank_words(cells);
  hipLaunchKernelGGL(idx_emit_kernel, dim3(grid_for(words)), dim3(256), 0, (hipStream_t)stream_,
                     static_cast<const uint2*>(index), words, g, out_indices);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_spconv_build_nbr(const void* in_index, const int32_t* in_perm, int batch, const int* in_shape,
                                    const int32_t* out_indices, int64_t m_out, const int* ksize, const int* stride,
                                    const int* pad, int32_t* nbr, void* stream_) {
  ConvGeom cg;
  if (int rc = make_conv(ksize, stride, pad, &cg)) return rc;
  Grid3 gin;
  unsigned long long cells;
  if (int rc = make_grid(batch, in_shape, &gin, &cells)) return rc;
  if (m_out == 0) return EFG_OK;
  hipLaunchKernelGGL(nbr_kernel, dim3(grid_for(m_out)), dim3(256), 0, (hipStream_t)stream_,
                     static_cast<const uint2*>(in_index), in_perm, gin, out_indices, (long long)m_out, cg, nbr);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_spconv_build_rnbr(const int32_t* nbr, int64_t m_out, int kvol, int64_t m_in, int32_t* rnbr,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(kvol >= 1 && m_out >= 0 && m_in >= 0, "spconv: bad rnbr sizes");
  if (m_in > 0) EFG_HIP_TRY(hipMemsetAsync(rnbr, 0xff, (size_t)m_in * kvol * 4, stream));
  if (m_out > 0 && m_in > 0) {
    hipLaunchKernelGGL(rnbr_kernel, dim3(grid_for(m_out * kvol)), dim3(256), 0, stream, nbr, (long long)m_out, kvol,
                       (long long)m_in, rnbr);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

extern "C" int efg_sparse_to_dense_f32(const float* feat, int c, const void* index, const int32_t* perm, int batch,
                                       const int* shape, float* dense, void* stream_) {
  Grid3 g;
  unsigned long long cells;
  if (int rc = make_grid(batch, shape, &g, &cells)) return rc;
  EFG_CHECK_ARG(c >= 1, "spconv: bad channel count");
  const long long dhw = (long long)shape[0] * shape[1] * shape[2];
  hipLaunchKernelGGL(to_dense_kernel, dim3((unsigned)ceil_div(dhw, 64), batch), dim3(256), 0, (hipStream_t)stream_,
                     feat, c, static_cast<const uint2*>(index), perm, dhw, dense);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_dense_to_sparse_f32(const float* grad_dense, int c, const int32_t* indices, int64_t m, int batch,
                                       const int* shape, float* grad_feat, void* stream_) {
  Grid3 g;
  unsigned long long cells;
  if (int rc = make_grid(batch, shape, &g, &cells)) return rc;
  EFG_CHECK_ARG(c >= 1 && m >= 0, "spconv: bad sizes");
  if (m == 0) return EFG_OK;
  hipLaunchKernelGGL(from_dense_kernel, dim3((unsigned)ceil_div(m, 64)), dim3(256), 0, (hipStream_t)stream_,
                     grad_dense, c, indices, (long long)m, g, grad_feat);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_sparse_to_bev_f32(const float* feat, int c, const void* index, const int32_t* perm, int batch,
                                     const int* shape, float* out, void* stream_) {
  Grid3 g;
  unsigned long long cells;
  if (int rc = make_grid(batch, shape, &g, &cells)) return rc;
  EFG_CHECK_ARG(c >= 1, "spconv: bad channel count");
  EFG_CHECK_ARG(shape[0] <= kBevMaxD, "sparse_to_bev: depth %d > %d not supported", shape[0], kBevMaxD);
  const long long npix = (long long)batch * shape[1] * shape[2];
  hipLaunchKernelGGL(to_bev_kernel, dim3((unsigned)ceil_div(npix, kBevPix)), dim3(256), 0, (hipStream_t)stream_, feat,
                     c, static_cast<const uint2*>(index), perm, g, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_bev_to_sparse_f32(const float* grad_out, int c, const int32_t* indices, int64_t m, int batch,
                                     const int* shape, float* grad_feat, void* stream_) {
  Grid3 g;
  unsigned long long cells;
  if (int rc = make_grid(batch, shape, &g, &cells)) return rc;
  EFG_CHECK_ARG(c >= 1 && m >= 0, "spconv: bad sizes");
  if (m == 0) return EFG_OK;
  hipLaunchKernelGGL(from_bev_kernel, dim3(grid_for(m * c)), dim3(256), 0, (hipStream_t)stream_, grad_out, c, indices,
                     (long long)m, g, grad_feat);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
