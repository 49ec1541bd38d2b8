// Box attention / multi-scale deformable attention sampling kernels for gfx950.
//
// One kernel family behind both efg::box_attn_forward/backward
// (efg/operators/src/box_attn/box_attn.h:29-83 -> box_attn_kernel.cuh:274-349 fwd, :352-472 bwd)
// and efg::ms_deform_attn_forward/backward (efg/operators/src/deform_attn/ms_deform_attn.h:22-63):
// the two reference families are the same math.
//
// Mapping (wave64-first, not a translation of the reference's thread-per-output-element /
// 32-thread-block layout):  a (batch, query, head) "pair" owns D contiguous channels.  It is
// served by LP = D/4 lanes, each holding one float4 of channels, so every bilinear corner is ONE
// 16-byte load per lane and a pair's corner is one contiguous 4*D-byte segment.  A wave carries
// 64/LP pairs (D = 32: the 8 heads of one query -> the wave's output row is 1 KiB contiguous).
// Sampling locations / weights are wave-broadcast loads.  Backward reduces grad_loc / grad_attn
// across the LP lanes with DPP shuffles (no LDS, no serial thread-0 sum) and accumulates
// grad_value with hardware fp32 atomics (global_atomic_add_f32).
#include "common.h"
#include <cmath>

namespace efg {
namespace {

struct MsdaDims {
  int b, s, h, d, l, lq, p;
  int lp;        // lanes per pair (power of two >= d/4)
  int lp_shift;  // log2(lp)
};

constexpr int kMaxLevels = 8;
struct Levels {
  int H[kMaxLevels], W[kMaxLevels];
  long long start[kMaxLevels];
};

// read spatial shapes / level starts (device int64) into registers
__device__ __forceinline__ void load_level(const long long* shapes, const long long* starts, int li, int& H, int& W,
                                           long long& st) {
  H = (int)shapes[li * 2];
  W = (int)shapes[li * 2 + 1];
  st = starts[li];
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float bil(float w1, float w2, float w3, float w4, float a, float b, float c, float d) {
  return fmaf(w4, d, fmaf(w3, c, fmaf(w2, b, w1 * a)));
}

// XCD-aware block remap: block b runs on XCD b % 8 (observed dispatch); give each XCD a contiguous
// range of pairs so that spatially adjacent queries share that XCD's L2 lines of `value`.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned per = nblk >> 3;
  if (per == 0 || bid >= (per << 3)) return bid;
  return (bid & 7) * per + (bid >> 3);
}

template <bool kBackward>
__global__ void __launch_bounds__(256)
msda_kernel(const float* __restrict__ value, const long long* __restrict__ shapes,
            const long long* __restrict__ starts, const float* __restrict__ loc, const float* __restrict__ attn,
            const float* __restrict__ grad_out, MsdaDims dm, float* __restrict__ out, float* __restrict__ grad_value,
            float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & 63;
  const int pairs_per_wave = 64 >> dm.lp_shift;
  const long long wave = (long long)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long t = wave * pairs_per_wave + (lane >> dm.lp_shift);  // (b*lq + q)*h + m
  const int c0 = (lane & (dm.lp - 1)) * 4;
  const long long total = (long long)dm.b * dm.lq * dm.h;
  const bool active = (t < total) && (c0 < dm.d);
  const long long tt = (t < total) ? t : 0;
  const int m = (int)(tt % dm.h);
  const int bi = (int)(tt / ((long long)dm.h * dm.lq));
  const int row_stride = dm.h * dm.d;  // floats between spatial positions
  const float* lw = loc + tt * dm.l * dm.p * 2;
  const float* aw = attn + tt * dm.l * dm.p;

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 top = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kBackward && active) top = ld4(grad_out + tt * dm.d + c0);

  for (int li = 0; li < dm.l; ++li) {
    int H, W;
    long long st;
    load_level(shapes, starts, li, H, W, st);
    const long long vbase = (((long long)bi * dm.s + st) * dm.h + m) * dm.d + c0;
    const float* v = value + vbase;
    float* gv = kBackward ? grad_value + vbase : nullptr;
    for (int pi = 0; pi < dm.p; ++pi) {
      const int e = li * dm.p + pi;
      const float loc_w = lw[e * 2], loc_h = lw[e * 2 + 1];
      const float wgt = aw[e];
      // pixel = loc * size - 0.5 (box_attn_kernel.cuh:322-323), two roundings like the reference
      const float h_im = __fsub_rn(__fmul_rn(loc_h, (float)H), 0.5f);
      const float w_im = __fsub_rn(__fmul_rn(loc_w, (float)W), 0.5f);
      const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
      float ga = 0.f, gw = 0.f, gh = 0.f;
      if (inside && active) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lwf;
        const float w1 = hh * hw, w2 = hh * lwf, w3 = lh * hw, w4 = lh * lwf;
        const bool t_ok = h_low >= 0, b_ok = h_high <= H - 1, l_ok = w_low >= 0, r_ok = w_high <= W - 1;
        const long long o1 = ((long long)h_low * W + w_low) * row_stride;
        const long long o2 = o1 + row_stride;
        const long long o3 = o1 + (long long)W * row_stride;
        const long long o4 = o3 + row_stride;
        // branch-free corner loads (clamped address, select afterwards): a predicated load becomes a branch
        // and the four loads then wait for each other
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const int y0 = max(h_low, 0), y1 = min(h_high, H - 1), x0 = max(w_low, 0), x1 = min(w_high, W - 1);
        const float4 u1 = ld4(v + ((long long)y0 * W + x0) * row_stride);
        const float4 u2 = ld4(v + ((long long)y0 * W + x1) * row_stride);
        const float4 u3 = ld4(v + ((long long)y1 * W + x0) * row_stride);
        const float4 u4 = ld4(v + ((long long)y1 * W + x1) * row_stride);
        const float4 v1 = (t_ok && l_ok) ? u1 : z;
        const float4 v2 = (t_ok && r_ok) ? u2 : z;
        const float4 v3 = (b_ok && l_ok) ? u3 : z;
        const float4 v4 = (b_ok && r_ok) ? u4 : z;
        float4 val;
        val.x = bil(w1, w2, w3, w4, v1.x, v2.x, v3.x, v4.x);
        val.y = bil(w1, w2, w3, w4, v1.y, v2.y, v3.y, v4.y);
        val.z = bil(w1, w2, w3, w4, v1.z, v2.z, v3.z, v4.z);
        val.w = bil(w1, w2, w3, w4, v1.w, v2.w, v3.w, v4.w);
        if (!kBackward) {
          acc.x = fmaf(val.x, wgt, acc.x);
          acc.y = fmaf(val.y, wgt, acc.y);
          acc.z = fmaf(val.z, wgt, acc.z);
          acc.w = fmaf(val.w, wgt, acc.w);
        } else {
          const float4 tv = make_float4(top.x * wgt, top.y * wgt, top.z * wgt, top.w * wgt);
          if (t_ok && l_ok) {
            unsafeAtomicAdd(gv + o1 + 0, w1 * tv.x); unsafeAtomicAdd(gv + o1 + 1, w1 * tv.y);
            unsafeAtomicAdd(gv + o1 + 2, w1 * tv.z); unsafeAtomicAdd(gv + o1 + 3, w1 * tv.w);
          }
          if (t_ok && r_ok) {
            unsafeAtomicAdd(gv + o2 + 0, w2 * tv.x); unsafeAtomicAdd(gv + o2 + 1, w2 * tv.y);
            unsafeAtomicAdd(gv + o2 + 2, w2 * tv.z); unsafeAtomicAdd(gv + o2 + 3, w2 * tv.w);
          }
          if (b_ok && l_ok) {
            unsafeAtomicAdd(gv + o3 + 0, w3 * tv.x); unsafeAtomicAdd(gv + o3 + 1, w3 * tv.y);
            unsafeAtomicAdd(gv + o3 + 2, w3 * tv.z); unsafeAtomicAdd(gv + o3 + 3, w3 * tv.w);
          }
          if (b_ok && r_ok) {
            unsafeAtomicAdd(gv + o4 + 0, w4 * tv.x); unsafeAtomicAdd(gv + o4 + 1, w4 * tv.y);
            unsafeAtomicAdd(gv + o4 + 2, w4 * tv.z); unsafeAtomicAdd(gv + o4 + 3, w4 * tv.w);
          }
          // d(out)/d(attn) = sum_c top_c * val_c ; d/d(loc) per box_attn_kernel.cuh:140-183
          ga = fmaf(top.w, val.w, fmaf(top.z, val.z, fmaf(top.y, val.y, top.x * val.x)));
          // grad_w_weight = -hh*v1 + hh*v2 - lh*v3 + lh*v4 ; grad_h_weight = -hw*v1 - lw*v2 + hw*v3 + lw*v4
          const float gwx = fmaf(hh, v2.x - v1.x, lh * (v4.x - v3.x)), ghx = fmaf(hw, v3.x - v1.x, lwf * (v4.x - v2.x));
          const float gwy = fmaf(hh, v2.y - v1.y, lh * (v4.y - v3.y)), ghy = fmaf(hw, v3.y - v1.y, lwf * (v4.y - v2.y));
          const float gwz = fmaf(hh, v2.z - v1.z, lh * (v4.z - v3.z)), ghz = fmaf(hw, v3.z - v1.z, lwf * (v4.z - v2.z));
          const float gww = fmaf(hh, v2.w - v1.w, lh * (v4.w - v3.w)), ghw = fmaf(hw, v3.w - v1.w, lwf * (v4.w - v2.w));
          gw = (float)W * fmaf(gww, tv.w, fmaf(gwz, tv.z, fmaf(gwy, tv.y, gwx * tv.x)));
          gh = (float)H * fmaf(ghw, tv.w, fmaf(ghz, tv.z, fmaf(ghy, tv.y, ghx * tv.x)));
        }
      }
      if (kBackward) {
        // reduce over the LP lanes of the pair (wave-uniform trip count; inactive lanes add 0)
        for (int dlt = dm.lp >> 1; dlt > 0; dlt >>= 1) {
          ga += __shfl_xor(ga, dlt, 64);
          gw += __shfl_xor(gw, dlt, 64);
          gh += __shfl_xor(gh, dlt, 64);
        }
        if (t < total && (lane & (dm.lp - 1)) == 0) {
          grad_attn[tt * dm.l * dm.p + e] = ga;
          grad_loc[(tt * dm.l * dm.p + e) * 2] = gw;
          grad_loc[(tt * dm.l * dm.p + e) * 2 + 1] = gh;
        }
      }
    }
  }
  if (!kBackward && active) *reinterpret_cast<float4*>(out + tt * dm.d + c0) = acc;
}

// ---- backward, grid-structured queries (encoder self-attention) -------------------------------
// When queries live on the value map itself (one level, Lq == H*W: box self-attention over the BEV
// tokens), query q at (qy,qx) samples a small box around itself, so the 4*P corner updates of
// neighbouring queries pile onto the same few cells: with global atomics that is ~10^9 contended
// L2 atomics per layer.  Here a workgroup owns a TQ x TQ tile of queries of ONE head and
// accumulates grad_value for the (TQ+2R)^2 window around the tile in LDS (fp64, ds_add_f64), then flushes
// the window once with global atomics (windows of neighbouring tiles overlap).  Corners that fall
// outside the window still go straight to global memory, so the result never depends on where the
// samples land -- only the speed does.
template <int D>
__global__ void __launch_bounds__(256)
msda_bwd_grid_kernel(const float* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
                     const float* __restrict__ grad_out, const long long* __restrict__ shapes, int nb, int nh,
                     long long s_total, int np, float* __restrict__ grad_value, float* __restrict__ grad_loc,
                     float* __restrict__ grad_attn) {
  constexpr int TQ = 8, R = 4, WIN = TQ + 2 * R, LP = D / 4;
  // the map shape lives on the device (int64 [1,2]); the grid is an upper bound over all H x W = S
  const int Hm = (int)shapes[0], Wm = (int)shapes[1];
  if ((long long)Hm * Wm != s_total) return;  // host dispatch guarantees this; defensive
  const int ntiles = ((Hm + TQ - 1) / TQ) * ((Wm + TQ - 1) / TQ);
  // the launch is sized for a square map and strides over the tiles: every H x W = S is covered without
  // a host read-back and without thousands of empty (64 KB LDS) workgroups
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  constexpr int SLOTS = 256 / LP;            // (query, head) pairs in flight per pass
  constexpr int PASSES = TQ * TQ / SLOTS;
  // fp64 accumulators: ds_add_f64 is native on gfx950 (3.1 lane-ops/clk/CU measured) while ds_add_f32
  // runs at 0.33 -- a 10x slower path (scripts/ubench/lds_atomics.hip)
  __shared__ double win[WIN * WIN * D];
  const int tiles_x = (Wm + TQ - 1) / TQ;
  const int ty0 = (tile / tiles_x) * TQ, tx0 = (tile % tiles_x) * TQ;
  const int wy0 = ty0 - R, wx0 = tx0 - R;
  const int m = blockIdx.y, bi = blockIdx.z;
  __syncthreads();  // previous tile's window flushed
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x % LP;          // float4 group of this lane
  const int slot = threadIdx.x / LP;
  const int c0 = sub * 4;
  const int rot = (slot + sub) & 3;          // spreads the 4 ds_add of a lane over the bank residues
  const long long S = (long long)Hm * Wm;
  const int row_stride = nh * D;
  for (int i = threadIdx.x; i < WIN * WIN * D; i += 256) win[i] = 0.0;
  __syncthreads();
  const float* v = value + ((long long)bi * S * nh + m) * D + c0;
  float* gv = grad_value + ((long long)bi * S * nh + m) * D + c0;
  for (int pass = 0; pass < PASSES; ++pass) {
    const int qi = pass * SLOTS + slot;
    const int qy = ty0 + qi / TQ, qx = tx0 + qi % TQ;
    const bool active = qy < Hm && qx < Wm;
    const long long t = active ? (((long long)bi * S + (long long)qy * Wm + qx) * nh + m) : 0;
    float4 top = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) top = ld4(grad_out + t * D + c0);
    const float* lw = loc + t * np * 2;
    const float* aw = attn + t * np;
    for (int pi = 0; pi < np; ++pi) {
      const float loc_w = lw[pi * 2], loc_h = lw[pi * 2 + 1];
      const float wgt = aw[pi];
      const float h_im = __fsub_rn(__fmul_rn(loc_h, (float)Hm), 0.5f);
      const float w_im = __fsub_rn(__fmul_rn(loc_w, (float)Wm), 0.5f);
      const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hm) && (w_im < (float)Wm);
      float ga = 0.f, gw = 0.f, gh = 0.f;
      if (inside && active) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lwf;
        const float wc[4] = {hh * hw, hh * lwf, lh * hw, lh * lwf};
        const float4 tv = make_float4(top.x * wgt, top.y * wgt, top.z * wgt, top.w * wgt);
        float4 vv[4];
#pragma unroll
        for (int cn = 0; cn < 4; ++cn) {  // branch-free loads: clamped address, select afterwards
          const int cy = min(max(h_low + (cn >> 1), 0), Hm - 1), cx = min(max(w_low + (cn & 1), 0), Wm - 1);
          vv[cn] = ld4(v + ((long long)cy * Wm + cx) * row_stride);
        }
#pragma unroll
        for (int cn = 0; cn < 4; ++cn) {
          const int cy = h_low + (cn >> 1), cx = w_low + (cn & 1);
          const bool ok = cy >= 0 && cy <= Hm - 1 && cx >= 0 && cx <= Wm - 1;
          if (!ok) vv[cn] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) {
            const long long o = ((long long)cy * Wm + cx) * row_stride;
            const float g[4] = {wc[cn] * tv.x, wc[cn] * tv.y, wc[cn] * tv.z, wc[cn] * tv.w};
            const int ly = cy - wy0, lx = cx - wx0;
            if ((unsigned)ly < (unsigned)WIN && (unsigned)lx < (unsigned)WIN) {
              double* wp = win + (ly * WIN + lx) * D + c0;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int jj = (j + rot) & 3;
                atomicAdd(wp + jj, (double)g[jj]);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) unsafeAtomicAdd(gv + o + j, g[j]);
            }
          }
        }
        float4 val;
        val.x = bil(wc[0], wc[1], wc[2], wc[3], vv[0].x, vv[1].x, vv[2].x, vv[3].x);
        val.y = bil(wc[0], wc[1], wc[2], wc[3], vv[0].y, vv[1].y, vv[2].y, vv[3].y);
        val.z = bil(wc[0], wc[1], wc[2], wc[3], vv[0].z, vv[1].z, vv[2].z, vv[3].z);
        val.w = bil(wc[0], wc[1], wc[2], wc[3], vv[0].w, vv[1].w, vv[2].w, vv[3].w);
        ga = fmaf(top.w, val.w, fmaf(top.z, val.z, fmaf(top.y, val.y, top.x * val.x)));
        const float gwx = fmaf(hh, vv[1].x - vv[0].x, lh * (vv[3].x - vv[2].x)), ghx = fmaf(hw, vv[2].x - vv[0].x, lwf * (vv[3].x - vv[1].x));
        const float gwy = fmaf(hh, vv[1].y - vv[0].y, lh * (vv[3].y - vv[2].y)), ghy = fmaf(hw, vv[2].y - vv[0].y, lwf * (vv[3].y - vv[1].y));
        const float gwz = fmaf(hh, vv[1].z - vv[0].z, lh * (vv[3].z - vv[2].z)), ghz = fmaf(hw, vv[2].z - vv[0].z, lwf * (vv[3].z - vv[1].z));
        const float gww = fmaf(hh, vv[1].w - vv[0].w, lh * (vv[3].w - vv[2].w)), ghw = fmaf(hw, vv[2].w - vv[0].w, lwf * (vv[3].w - vv[1].w));
        gw = (float)Wm * fmaf(gww, tv.w, fmaf(gwz, tv.z, fmaf(gwy, tv.y, gwx * tv.x)));
        gh = (float)Hm * fmaf(ghw, tv.w, fmaf(ghz, tv.z, fmaf(ghy, tv.y, ghx * tv.x)));
      }
#pragma unroll
      for (int dlt = LP >> 1; dlt > 0; dlt >>= 1) {
        ga += __shfl_xor(ga, dlt, 64);
        gw += __shfl_xor(gw, dlt, 64);
        gh += __shfl_xor(gh, dlt, 64);
      }
      if (active && sub == 0) {
        grad_attn[t * np + pi] = ga;
        grad_loc[(t * np + pi) * 2] = gw;
        grad_loc[(t * np + pi) * 2 + 1] = gh;
      }
    }
  }
  (void)lane;
  __syncthreads();
  // flush the window: one thread per (cell, channel); skip untouched entries
  for (int i = threadIdx.x; i < WIN * WIN * D; i += 256) {
    const float g = (float)win[i];
    if (g != 0.0f) {
      const int cell = i / D, ch = i % D;
      const int cy = wy0 + cell / WIN, cx = wx0 + cell % WIN;
      if (cy >= 0 && cy < Hm && cx >= 0 && cx < Wm)
        unsafeAtomicAdd(grad_value + (((long long)bi * S + (long long)cy * Wm + cx) * nh + m) * D + ch, g);
    }
  }
  }  // tile loop
}

int check_dims(int b, int s, int h, int d, int l, int lq, int p, MsdaDims* dm) {
  EFG_CHECK_ARG(b >= 0 && s >= 0 && h >= 1 && l >= 1 && lq >= 0 && p >= 1, "msda: bad dimensions");
  EFG_CHECK_ARG(d >= 4 && d % 4 == 0 && d <= 256, "msda: head dim must be a multiple of 4 in [4,256], got %d", d);
  EFG_CHECK_ARG(l <= kMaxLevels, "msda: at most %d levels", kMaxLevels);
  int lp = 1, sh = 0;
  while (lp * 4 < d) {
    lp <<= 1;
    ++sh;
  }
  *dm = MsdaDims{b, s, h, d, l, lq, p, lp, sh};
  return EFG_OK;
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_msda_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                    const float* loc, const float* attn, int b, int s, int h, int d, int l, int lq,
                                    int p, float* out, void* stream) {
  MsdaDims dm;
  if (int rc = check_dims(b, s, h, d, l, lq, p, &dm)) return rc;
  const long long total = (long long)b * lq * h;
  if (total == 0) return EFG_OK;
  const int pairs_per_block = 4 * (64 / dm.lp);
  const unsigned blocks = (unsigned)ceil_div(total, pairs_per_block);
  hipLaunchKernelGGL((msda_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, value,
                     (const long long*)shapes, (const long long*)level_start, loc, attn, (const float*)nullptr, dm,
                     out, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_msda_backward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                     const float* loc, const float* attn, const float* grad_out, int b, int s, int h,
                                     int d, int l, int lq, int p, float* grad_value, float* grad_loc,
                                     float* grad_attn, void* stream) {
  MsdaDims dm;
  if (int rc = check_dims(b, s, h, d, l, lq, p, &dm)) return rc;
  const long long total = (long long)b * lq * h;
  if (total == 0) return EFG_OK;
  if (l == 1 && d == 32 && s == lq && s >= 1024 && h <= 65535 && b <= 65535) {
    // queries on the value grid (encoder self-attention): LDS-window accumulation.  H and W are
    // device-side (int64): the launch is sized for a square map and the kernel strides over the tiles.
    const int side = (int)std::ceil(std::sqrt((double)s));
    const unsigned tiles_sq = (unsigned)(((side + 7) / 8) * ((side + 7) / 8));
    hipLaunchKernelGGL((msda_bwd_grid_kernel<32>), dim3(tiles_sq, h, b), dim3(256), 0, (hipStream_t)stream, value, loc,
                       attn, grad_out, (const long long*)shapes, b, h, (long long)s, p, grad_value, grad_loc,
                       grad_attn);
    EFG_LAUNCH_CHECK();
    return EFG_OK;
  }
  const int pairs_per_block = 4 * (64 / dm.lp);
  const unsigned blocks = (unsigned)ceil_div(total, pairs_per_block);
  hipLaunchKernelGGL((msda_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, value,
                     (const long long*)shapes, (const long long*)level_start, loc, attn, grad_out, dm,
                     (float*)nullptr, grad_value, grad_loc, grad_attn);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
