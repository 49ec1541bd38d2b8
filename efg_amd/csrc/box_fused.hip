// Fused Box3dAttention sampling for gfx950: box geometry + softmax + bilinear sampling in ONE kernel.
//
// The reference evaluates Box3dAttention.forward ($CQ/modules/box_attention.py:62-115) as ~20 PyTorch
// elementwise kernels that materialise the [B, Lq, H, L, 25, 2] sampling grid (85 MB per scene per
// encoder layer), a [.., 25, 2, 2] rotation product (2x that) and the softmaxed weights, and then calls
// BoxAttnFunction on them.  Here the kernel reads only what is independent per (query, head):
//   ref window [7], box offsets [V = 4|5], attention logits [L*P], the P x 2 kernel lattice,
// and rebuilds centre / size / rotation / softmax in registers.  Backward returns gradients w.r.t. the
// value map, the raw box offsets and the raw logits (softmax and geometry backward fused in).
// Same lane mapping as msda.hip (LP = D/4 lanes x float4 per pair); encoder self-attention
// (queries on the value grid) accumulates grad_value in an fp64 LDS window like msda_bwd_grid_kernel.
#include "common.h"
#include <cmath>

namespace efg {
namespace {

constexpr int kMaxLevels = 8;
constexpr int kMaxPts = 128;  // L * P upper bound for the fused path (LDS scratch)
constexpr int kSoftmaxLds = 4096;  // floats: (pairs per workgroup) x (L*P) must fit

struct BoxDims {
  int b, s, h, d, l, lq, p, v;  // v = 4 (no rotation) or 5
  int lp, lp_shift;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float bil(float w1, float w2, float w3, float w4, float a, float b, float c, float d) {
  return fmaf(w4, d, fmaf(w3, c, fmaf(w2, b, w1 * a)));
}

struct BoxGeo {
  float cx, cy, w, h, cs, sn;   // centre, relu'd size, cos / sin
  float rw, rh;                 // reference size (chain rule of the offsets)
  bool w_on, h_on;              // relu masks
};

// boxes = ref[x,y,l,w] + off/8 * ref[l,w,l,w]; angle = (ref_a + off_a/16) * 2pi (rotation) or ref_a
__device__ __forceinline__ BoxGeo make_box(const float* __restrict__ ref, const float* __restrict__ off, int v) {
  BoxGeo g;
  const float rx = ref[0], ry = ref[1];
  g.rw = ref[3];
  g.rh = ref[4];
  g.cx = rx + off[0] / 8.0f * g.rw;
  g.cy = ry + off[1] / 8.0f * g.rh;
  const float w = g.rw + off[2] / 8.0f * g.rw, h = g.rh + off[3] / 8.0f * g.rh;
  g.w_on = w > 0.0f;
  g.h_on = h > 0.0f;
  g.w = g.w_on ? w : 0.0f;
  g.h = g.h_on ? h : 0.0f;
  const float ang = (v == 5) ? (ref[6] + off[4] / 16.0f) * 2.0f * 3.14159274f : ref[6];
  g.cs = cosf(ang);
  g.sn = sinf(ang);
  return g;
}

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned per = nblk >> 3;
  if (per == 0 || bid >= (per << 3)) return bid;
  return (bid & 7) * per + (bid >> 3);
}

// ---- forward ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
box_fwd_kernel(const float* __restrict__ value, const long long* __restrict__ shapes,
               const long long* __restrict__ starts, const float* __restrict__ ref, const float* __restrict__ off,
               const float* __restrict__ logits, const float* __restrict__ kidx, BoxDims dm,
               float* __restrict__ out) {
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & 63;
  const int pairs_per_wave = 64 >> dm.lp_shift;
  const long long wave = (long long)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long t = wave * pairs_per_wave + (lane >> dm.lp_shift);
  const int c0 = (lane & (dm.lp - 1)) * 4;
  const long long total = (long long)dm.b * dm.lq * dm.h;
  const bool active = (t < total) && (c0 < dm.d);
  const long long tt = (t < total) ? t : 0;
  const int m = (int)(tt % dm.h);
  const long long bq = tt / dm.h;
  const int bi = (int)(bq / dm.lq);
  const int row_stride = dm.h * dm.d;
  const int np = dm.l * dm.p;
  const float* lg = logits + tt * np;
  // softmax over the L*P logits of the pair, shared by its LP lanes: each lane exponentiates every LP-th
  // logit, the weights go through LDS (the redundant form costs 2*L*P expf per lane)
  __shared__ float a_s[kSoftmaxLds];
  const int sub = lane & (dm.lp - 1);
  float* as = a_s + (threadIdx.x >> dm.lp_shift) * np;
  float mx = -INFINITY;
  for (int e = sub; e < np; e += dm.lp) mx = fmaxf(mx, lg[e]);
  for (int dlt = dm.lp >> 1; dlt > 0; dlt >>= 1) mx = fmaxf(mx, __shfl_xor(mx, dlt, 64));
  float den = 0.0f;
  for (int e = sub; e < np; e += dm.lp) {
    const float ex = expf(lg[e] - mx);
    as[e] = ex;
    den += ex;
  }
  for (int dlt = dm.lp >> 1; dlt > 0; dlt >>= 1) den += __shfl_xor(den, dlt, 64);
  const float inv = 1.0f / den;
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int li = 0; li < dm.l; ++li) {
    const int H = (int)shapes[li * 2], W = (int)shapes[li * 2 + 1];
    const float* v = value + (((long long)bi * dm.s + starts[li]) * dm.h + m) * dm.d + c0;
    const BoxGeo g = make_box(ref + bq * 7, off + (tt * dm.l + li) * dm.v, dm.v);
    for (int pi = 0; pi < dm.p; ++pi) {
      const float gx = kidx[pi * 2] * g.w, gy = kidx[pi * 2 + 1] * g.h;
      const float loc_w = g.cx + (gx * g.cs + gy * (-g.sn));
      const float loc_h = g.cy + (gx * g.sn + gy * g.cs);
      const float wgt = as[li * dm.p + pi] * inv;
      const float h_im = __fsub_rn(__fmul_rn(loc_h, (float)H), 0.5f);
      const float w_im = __fsub_rn(__fmul_rn(loc_w, (float)W), 0.5f);
      const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
      if (inside && active) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lwf;
        const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
        // branch-free corner loads: clamp the address, always load, zero the WEIGHT of an outside corner
        // (a predicated load becomes a branch and the four loads then wait for each other)
        const int y0 = max(h_low, 0), y1 = min(h_low + 1, H - 1), x0 = max(w_low, 0), x1 = min(w_low + 1, W - 1);
        const float4 v1 = ld4(v + ((long long)y0 * W + x0) * row_stride);
        const float4 v2 = ld4(v + ((long long)y0 * W + x1) * row_stride);
        const float4 v3 = ld4(v + ((long long)y1 * W + x0) * row_stride);
        const float4 v4 = ld4(v + ((long long)y1 * W + x1) * row_stride);
        const float w1 = (t_ok && l_ok) ? hh * hw : 0.f, w2 = (t_ok && r_ok) ? hh * lwf : 0.f;
        const float w3 = (b_ok && l_ok) ? lh * hw : 0.f, w4 = (b_ok && r_ok) ? lh * lwf : 0.f;
        acc.x = fmaf(bil(w1, w2, w3, w4, v1.x, v2.x, v3.x, v4.x), wgt, acc.x);
        acc.y = fmaf(bil(w1, w2, w3, w4, v1.y, v2.y, v3.y, v4.y), wgt, acc.y);
        acc.z = fmaf(bil(w1, w2, w3, w4, v1.z, v2.z, v3.z, v4.z), wgt, acc.z);
        acc.w = fmaf(bil(w1, w2, w3, w4, v1.w, v2.w, v3.w, v4.w), wgt, acc.w);
      }
    }
  }
  if (active) *reinterpret_cast<float4*>(out + tt * dm.d + c0) = acc;
}

// ---- backward -----------------------------------------------------------------------------------
// One (query, head) pair per LP-lane group.  kWin: queries are the cells of a single-level value map
// (encoder self-attention); the workgroup then owns an 8x8 query tile of one head and accumulates
// grad_value for the 16x16 window around it in fp64 LDS (see msda.hip for the measurements).
template <int D, bool kWin, int PTS = kMaxPts>
__global__ void __launch_bounds__(256)
box_bwd_kernel(const float* __restrict__ value, const long long* __restrict__ shapes,
               const long long* __restrict__ starts, const float* __restrict__ ref, const float* __restrict__ off,
               const float* __restrict__ logits, const float* __restrict__ kidx, const float* __restrict__ grad_out,
               BoxDims dm, float* __restrict__ grad_value, float* __restrict__ grad_off,
               float* __restrict__ grad_logits) {
  constexpr int LP = D / 4, SLOTS = 256 / LP;
  constexpr int TQ = 8, R = 4, WIN = TQ + 2 * R;
  __shared__ double win[kWin ? WIN * WIN * D : 1];
  __shared__ float ga_s[SLOTS][PTS];
  __shared__ float a_s[SLOTS][PTS];  // un-normalised softmax weights of the pair
  const int sub = threadIdx.x % LP, slot = threadIdx.x / LP;
  const int c0 = sub * 4;
  const int rot = (slot + sub) & 3;
  const int np = dm.l * dm.p;
  const int row_stride = dm.h * dm.d;
  int Hm = 0, Wm = 0, wy0 = 0, wx0 = 0, ty0 = 0, tx0 = 0, m_fixed = 0, bi_fixed = 0;
  int passes = 1;
  if (kWin) {
    Hm = (int)shapes[0];
    Wm = (int)shapes[1];
    if ((long long)Hm * Wm != dm.s) return;
    m_fixed = blockIdx.y;
    bi_fixed = blockIdx.z;
    passes = TQ * TQ / SLOTS;
  }
  const int tiles_x = kWin ? (Wm + TQ - 1) / TQ : 1;
  const int ntiles = kWin ? ((Hm + TQ - 1) / TQ) * tiles_x : 1;
  for (int tile = kWin ? blockIdx.x : 0; tile < ntiles; tile += kWin ? gridDim.x : 1) {
  if (kWin) {
    ty0 = (tile / tiles_x) * TQ;
    tx0 = (tile % tiles_x) * TQ;
    wy0 = ty0 - R;
    wx0 = tx0 - R;
    __syncthreads();  // previous tile's window flushed
    for (int i = threadIdx.x; i < WIN * WIN * D; i += 256) win[i] = 0.0;
    __syncthreads();
  }
  const long long total = (long long)dm.b * dm.lq * dm.h;
  for (int pass = 0; pass < passes; ++pass) {
    long long t;
    bool qok;
    if (kWin) {
      const int qi = pass * SLOTS + slot;
      const int qy = ty0 + qi / TQ, qx = tx0 + qi % TQ;
      qok = qy < Hm && qx < Wm;
      t = qok ? (((long long)bi_fixed * dm.lq + (long long)qy * Wm + qx) * dm.h + m_fixed) : 0;
    } else {
      const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
      t = (long long)bid * SLOTS + slot;
      qok = t < total;
      if (!qok) t = 0;
    }
    const bool active = qok && c0 < dm.d;
    const int m = (int)(t % dm.h);
    const long long bq = t / dm.h;
    const int bi = (int)(bq / dm.lq);
    const float* lg = logits + t * np;
    float* as = a_s[slot];
    float mx = -INFINITY;
    for (int e = sub; e < np; e += LP) mx = fmaxf(mx, lg[e]);
#pragma unroll
    for (int dlt = LP >> 1; dlt > 0; dlt >>= 1) mx = fmaxf(mx, __shfl_xor(mx, dlt, 64));
    float den = 0.0f;
    for (int e = sub; e < np; e += LP) {
      const float ex = expf(lg[e] - mx);
      as[e] = ex;
      den += ex;
    }
#pragma unroll
    for (int dlt = LP >> 1; dlt > 0; dlt >>= 1) den += __shfl_xor(den, dlt, 64);
    const float inv = 1.0f / den;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float4 top = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) top = ld4(grad_out + t * dm.d + c0);
    float dot = 0.0f;  // sum_p a_p * ga_p (softmax backward)
    for (int li = 0; li < dm.l; ++li) {
      const int H = (int)shapes[li * 2], W = (int)shapes[li * 2 + 1];
      const long long vbase = (((long long)bi * dm.s + starts[li]) * dm.h + m) * dm.d + c0;
      const float* v = value + vbase;
      float* gv = grad_value + vbase;
      const BoxGeo g = make_box(ref + bq * 7, off + (t * dm.l + li) * dm.v, dm.v);
      float dcx = 0.f, dcy = 0.f, dw = 0.f, dh = 0.f, dth = 0.f;
      for (int pi = 0; pi < dm.p; ++pi) {
        const float kxn = kidx[pi * 2], kyn = kidx[pi * 2 + 1];
        const float gx = kxn * g.w, gy = kyn * g.h;
        const float loc_w = g.cx + (gx * g.cs + gy * (-g.sn));
        const float loc_h = g.cy + (gx * g.sn + gy * g.cs);
        const float wgt = as[li * dm.p + pi] * inv;
        const float h_im = __fsub_rn(__fmul_rn(loc_h, (float)H), 0.5f);
        const float w_im = __fsub_rn(__fmul_rn(loc_w, (float)W), 0.5f);
        const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
        float ga = 0.f, gwl = 0.f, ghl = 0.f;
        if (inside && active) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
          const float hh = 1.f - lh, hw = 1.f - lwf;
          const float wc[4] = {hh * hw, hh * lwf, lh * hw, lh * lwf};
          const float4 tv = make_float4(top.x * wgt, top.y * wgt, top.z * wgt, top.w * wgt);
          float4 vv[4];
#pragma unroll
          for (int cn = 0; cn < 4; ++cn) {  // branch-free loads: clamped address, select afterwards
            const int cy = min(max(h_low + (cn >> 1), 0), H - 1), cx = min(max(w_low + (cn & 1), 0), W - 1);
            vv[cn] = ld4(v + ((long long)cy * W + cx) * row_stride);
          }
#pragma unroll
          for (int cn = 0; cn < 4; ++cn) {
            const int cy = h_low + (cn >> 1), cx = w_low + (cn & 1);
            const bool ok = cy >= 0 && cy <= H - 1 && cx >= 0 && cx <= W - 1;
            if (!ok) vv[cn] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
              const long long o = ((long long)cy * W + cx) * row_stride;
              const float gq[4] = {wc[cn] * tv.x, wc[cn] * tv.y, wc[cn] * tv.z, wc[cn] * tv.w};
              const int ly = cy - wy0, lx = cx - wx0;
              if (kWin && (unsigned)ly < (unsigned)WIN && (unsigned)lx < (unsigned)WIN) {
                double* wp = win + (ly * WIN + lx) * D + c0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int jj = (j + rot) & 3;
                  atomicAdd(wp + jj, (double)gq[jj]);
                }
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(gv + o + j, gq[j]);
              }
            }
          }
          const float vx = bil(wc[0], wc[1], wc[2], wc[3], vv[0].x, vv[1].x, vv[2].x, vv[3].x);
          const float vy = bil(wc[0], wc[1], wc[2], wc[3], vv[0].y, vv[1].y, vv[2].y, vv[3].y);
          const float vz = bil(wc[0], wc[1], wc[2], wc[3], vv[0].z, vv[1].z, vv[2].z, vv[3].z);
          const float vw = bil(wc[0], wc[1], wc[2], wc[3], vv[0].w, vv[1].w, vv[2].w, vv[3].w);
          ga = fmaf(top.w, vw, fmaf(top.z, vz, fmaf(top.y, vy, top.x * vx)));
          const float gwx = fmaf(hh, vv[1].x - vv[0].x, lh * (vv[3].x - vv[2].x)), ghx = fmaf(hw, vv[2].x - vv[0].x, lwf * (vv[3].x - vv[1].x));
          const float gwy = fmaf(hh, vv[1].y - vv[0].y, lh * (vv[3].y - vv[2].y)), ghy = fmaf(hw, vv[2].y - vv[0].y, lwf * (vv[3].y - vv[1].y));
          const float gwz = fmaf(hh, vv[1].z - vv[0].z, lh * (vv[3].z - vv[2].z)), ghz = fmaf(hw, vv[2].z - vv[0].z, lwf * (vv[3].z - vv[1].z));
          const float gww = fmaf(hh, vv[1].w - vv[0].w, lh * (vv[3].w - vv[2].w)), ghw = fmaf(hw, vv[2].w - vv[0].w, lwf * (vv[3].w - vv[1].w));
          gwl = (float)W * fmaf(gww, tv.w, fmaf(gwz, tv.z, fmaf(gwy, tv.y, gwx * tv.x)));
          ghl = (float)H * fmaf(ghw, tv.w, fmaf(ghz, tv.z, fmaf(ghy, tv.y, ghx * tv.x)));
        }
#pragma unroll
        for (int dlt = LP >> 1; dlt > 0; dlt >>= 1) ga += __shfl_xor(ga, dlt, 64);
        // chain rule through grid = centre + R(theta) . (k * size); gwl / ghl are this lane's share (its 4
        // channels) -- the sums over points stay lane-partial and are reduced across the LP lanes once
        dcx += gwl;
        dcy += ghl;
        dw += kxn * (gwl * g.cs + ghl * g.sn);
        dh += kyn * (ghl * g.cs - gwl * g.sn);
        dth += gwl * (-(gx * g.sn) - gy * g.cs) + ghl * (gx * g.cs - gy * g.sn);
        dot = fmaf(wgt, ga, dot);
        if (sub == 0) ga_s[slot][li * dm.p + pi] = ga;
      }
#pragma unroll
      for (int dlt = LP >> 1; dlt > 0; dlt >>= 1) {
        dcx += __shfl_xor(dcx, dlt, 64);
        dcy += __shfl_xor(dcy, dlt, 64);
        dw += __shfl_xor(dw, dlt, 64);
        dh += __shfl_xor(dh, dlt, 64);
        dth += __shfl_xor(dth, dlt, 64);
      }
      if (qok && sub == 0) {
        float* go = grad_off + (t * dm.l + li) * dm.v;
        go[0] = dcx * g.rw / 8.0f;
        go[1] = dcy * g.rh / 8.0f;
        go[2] = g.w_on ? dw * g.rw / 8.0f : 0.0f;
        go[3] = g.h_on ? dh * g.rh / 8.0f : 0.0f;
        if (dm.v == 5) go[4] = dth * (2.0f * 3.14159274f / 16.0f);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // softmax backward: d logit_p = a_p * (ga_p - sum_j a_j ga_j); the LP lanes of the pair share the work
    if (qok) {
      for (int e = sub; e < np; e += LP) grad_logits[t * np + e] = as[e] * inv * (ga_s[slot][e] - dot);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (kWin) {
    __syncthreads();
    const long long S = dm.s;
    for (int i = threadIdx.x; i < WIN * WIN * D; i += 256) {
      const float gq = (float)win[i];
      if (gq != 0.0f) {
        const int cell = i / D, ch = i % D;
        const int cy = wy0 + cell / WIN, cx = wx0 + cell % WIN;
        if (cy >= 0 && cy < Hm && cx >= 0 && cx < Wm)
          unsafeAtomicAdd(grad_value + (((long long)bi_fixed * S + (long long)cy * Wm + cx) * dm.h + m_fixed) * D + ch, gq);
      }
    }
  }
  }  // tile loop
}

int check(int b, int s, int h, int d, int l, int lq, int p, int v, BoxDims* dm) {
  EFG_CHECK_ARG(b >= 0 && s >= 0 && h >= 1 && l >= 1 && lq >= 0 && p >= 1, "box_attn_fused: bad dimensions");
  EFG_CHECK_ARG(v == 4 || v == 5, "box_attn_fused: offsets must have 4 or 5 variables, got %d", v);
  EFG_CHECK_ARG(l <= kMaxLevels && l * p <= kMaxPts, "box_attn_fused: at most %d levels and %d points in total", kMaxLevels, kMaxPts);
  EFG_CHECK_ARG(d >= 4 && d % 4 == 0 && d <= 256, "box_attn_fused: head dim must be a multiple of 4 in [4,256]");
  int lp = 1, sh = 0;
  while (lp * 4 < d) {
    lp <<= 1;
    ++sh;
  }
  EFG_CHECK_ARG((256 / lp) * l * p <= kSoftmaxLds, "box_attn_fused: (256/(d/4)) * L*P = %d exceeds the LDS scratch (%d)",
                (256 / lp) * l * p, kSoftmaxLds);
  *dm = BoxDims{b, s, h, d, l, lq, p, v, lp, sh};
  return EFG_OK;
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_box_attn_fused_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                              const float* ref_windows, const float* offsets, const float* logits,
                                              const float* kernel_indices, int b, int s, int h, int d, int l, int lq,
                                              int p, int v, float* out, void* stream) {
  BoxDims dm;
  if (int rc = check(b, s, h, d, l, lq, p, v, &dm)) return rc;
  const long long total = (long long)b * lq * h;
  if (total == 0) return EFG_OK;
  const int pairs_per_block = 4 * (64 / dm.lp);
  hipLaunchKernelGGL(box_fwd_kernel, dim3((unsigned)ceil_div(total, pairs_per_block)), dim3(256), 0,
                     (hipStream_t)stream, value, (const long long*)shapes, (const long long*)level_start, ref_windows,
                     offsets, logits, kernel_indices, dm, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_box_attn_fused_backward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                               const float* ref_windows, const float* offsets, const float* logits,
                                               const float* kernel_indices, const float* grad_out, int b, int s, int h,
                                               int d, int l, int lq, int p, int v, float* grad_value,
                                               float* grad_offsets, float* grad_logits, void* stream) {
  BoxDims dm;
  if (int rc = check(b, s, h, d, l, lq, p, v, &dm)) return rc;
  EFG_CHECK_ARG(d == 32, "box_attn_fused backward: head dim 32 only (got %d); use the unfused op", d);
  const long long total = (long long)b * lq * h;
  if (total == 0) return EFG_OK;
  if (l == 1 && s == lq && s >= 1024 && h <= 65535 && b <= 65535) {
    // encoder self-attention (queries on the value map): fp64 LDS window per 8x8 query tile.  H, W are
    // device-side, so the launch is sized for a square map and the kernel strides over the tiles.
    const int side = (int)std::ceil(std::sqrt((double)s));
    const unsigned tiles_sq = (unsigned)(((side + 7) / 8) * ((side + 7) / 8));
    if (l * p <= 32)
      hipLaunchKernelGGL((box_bwd_kernel<32, true, 32>), dim3(tiles_sq, h, b), dim3(256), 0, (hipStream_t)stream,
                         value, (const long long*)shapes, (const long long*)level_start, ref_windows, offsets, logits,
                         kernel_indices, grad_out, dm, grad_value, grad_offsets, grad_logits);
    else
      hipLaunchKernelGGL((box_bwd_kernel<32, true>), dim3(tiles_sq, h, b), dim3(256), 0, (hipStream_t)stream,
                         value, (const long long*)shapes, (const long long*)level_start, ref_windows, offsets, logits,
                         kernel_indices, grad_out, dm, grad_value, grad_offsets, grad_logits);
  } else {
    hipLaunchKernelGGL((box_bwd_kernel<32, false>), dim3((unsigned)ceil_div(total, 32)), dim3(256), 0,
                       (hipStream_t)stream, value, (const long long*)shapes, (const long long*)level_start,
                       ref_windows, offsets, logits, kernel_indices, grad_out, dm, grad_value, grad_offsets,
                       grad_logits);
  }
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
