// Fused Box3dAttention sampling for gfx950: box geometry + softmax + bilinear sampling in ONE kernel.
//
// The reference evaluates Box3dAttention.forward ($CQ/modules/box_attention.py:62-115) as ~20 PyTorch
// elementwise kernels that materialise the [B, Lq, H, L, 25, 2] sampling grid (85 MB per scene per
// encoder layer), a [.., 25, 2, 2] rotation product (2x that) and the softmaxed weights, and then calls
// BoxAttnFunction on them.  Here the kernel reads only what is independent per (query, head):
//   ref window [7], box offsets [V = 4|5], attention logits [L*P], the P x 2 kernel lattice,
// and rebuilds centre / size / rotation / softmax in registers.  Backward returns gradients w.r.t. the
// value map, the raw box offsets and the raw logits (softmax and geometry backward fused in).
// Same lane mapping as msda.hip (LP = D/4 lanes x float4 per pair).  Encoder self-attention (queries on the
// value grid, L*P <= 32) has its own backward in GEMM form, box_bwd_tile_kernel below; larger point counts
// accumulate grad_value in an fp64 LDS window like msda_bwd_grid_kernel.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace efg {
namespace {

constexpr int kMaxLevels = 8;
constexpr int kAbsmaxSlots = 63;   // words 1..63 of the bin workspace's flag block: partial maxima of |grad_out| (absmax_bits_kernel)
constexpr int kMaxPts = 128;  // L * P upper bound for the fused path (LDS scratch)
constexpr int kSoftmaxLds = 4096;  // floats: (pairs per workgroup) x (L*P) must fit

struct BoxDims {
  int b, s, h, d, l, lq, p, v;  // v = 4 (no rotation) or 5
  int lp, lp_shift;
  // row strides (floats) of the offsets / logits matrices and of their gradients: h * l * v and h * l * p when each is a
  // matrix of its own, the width of the shared matrix when both come out of ONE projection (Box3dAttention)
  int off_rs, lg_rs;
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

// boxes = ref[x,y,l,w] + off/8 * ref[l,w,l,w]; angle = (ref_a + off_a/16) * 2pi (rotation) or ref_a.
// ONE definition of the geometry for every kernel of this file, written with explicit IEEE roundings (no FMA
// contraction, which the compiler may apply differently per kernel): box_bin_count_kernel and the backward
// kernels must put every corner into the same cell bit for bit, or a bin's entries run into its neighbour's.
// (The reference evaluates the same expressions as separate PyTorch elementwise ops, i.e. without FMA too.)
__device__ __forceinline__ BoxGeo make_box_v(const float rf[5], const float of[5], int v) {
  BoxGeo g;
  g.rw = rf[2];
  g.rh = rf[3];
  g.cx = __fadd_rn(rf[0], __fmul_rn(__fdiv_rn(of[0], 8.0f), g.rw));
  g.cy = __fadd_rn(rf[1], __fmul_rn(__fdiv_rn(of[1], 8.0f), g.rh));
  const float w = __fadd_rn(g.rw, __fmul_rn(__fdiv_rn(of[2], 8.0f), g.rw));
  const float h = __fadd_rn(g.rh, __fmul_rn(__fdiv_rn(of[3], 8.0f), g.rh));
  g.w_on = w > 0.0f;
  g.h_on = h > 0.0f;
  g.w = g.w_on ? w : 0.0f;
  g.h = g.h_on ? h : 0.0f;
  const float ang = (v == 5) ? __fmul_rn(__fmul_rn(__fadd_rn(rf[4], __fdiv_rn(of[4], 16.0f)), 2.0f), 3.14159274f) : rf[4];
  if (ang == 0.0f) {   // axis-aligned anchors (the encoder): cosf(0) = 1 and sinf(0) = 0 exactly
    g.cs = 1.0f;
    g.sn = 0.0f;
  } else {
    g.cs = cosf(ang);
    g.sn = sinf(ang);
  }
  return g;
}

__device__ __forceinline__ BoxGeo make_box(const float* __restrict__ ref, const float* __restrict__ off, int v) {
  const float rf[5] = {ref[0], ref[1], ref[3], ref[4], ref[6]};
  const float of[5] = {off[0], off[1], off[2], off[3], v == 5 ? off[4] : 0.0f};
  return make_box_v(rf, of, v);
}

// lattice point (kxn, kyn) of box g on an H x W map: offsets from the centre (gx, gy, in box-normalised units) and
// the pixel position (h_im, w_im) = loc * size - 0.5
struct BoxPx {
  float gx, gy, h_im, w_im;
};
__device__ __forceinline__ BoxPx box_point(const BoxGeo& g, float kxn, float kyn, int H, int W) {
  BoxPx p;
  p.gx = __fmul_rn(kxn, g.w);
  p.gy = __fmul_rn(kyn, g.h);
  const float loc_w = __fadd_rn(g.cx, __fsub_rn(__fmul_rn(p.gx, g.cs), __fmul_rn(p.gy, g.sn)));
  const float loc_h = __fadd_rn(g.cy, __fadd_rn(__fmul_rn(p.gx, g.sn), __fmul_rn(p.gy, g.cs)));
  p.h_im = __fsub_rn(__fmul_rn(loc_h, (float)H), 0.5f);
  p.w_im = __fsub_rn(__fmul_rn(loc_w, (float)W), 0.5f);
  return p;
}

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned per = nblk >> 3;
  if (per == 0 || bid >= (per << 3)) return bid;
  return (bid & 7) * per + (bid >> 3);
}

// One (grad_out row, weight) entry into `bin`.  cursor[bin] was initialised to the bin's first slot and bin_end[bin]
// is the first slot of the next bin (box_bin_count_kernel counted with the same geometry, so the slot is always
// inside); an entry that does not fit is dropped and counted in *overflow, which the host-side tests read -- it
// must never overwrite a neighbour's entries.
__device__ __forceinline__ void bin_push(int* __restrict__ cursor, const int* __restrict__ bin_end,
                                         int* __restrict__ overflow, int2* __restrict__ entries, long long bin,
                                         int row, float w) {
  const int slot = atomicAdd(cursor + bin, 1);
  if (slot < bin_end[bin])
    entries[slot] = make_int2(row, __float_as_int(w));
  else
    atomicAdd(overflow, 1);
}

// ---- DPP helpers: the 4 lanes of a quad ------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// lane k of the quad, k a compile-time constant after unrolling
__device__ __forceinline__ int quad_bcast_i(int v, int k) {
  switch (k) {
    case 0: return quad_perm_i<0x00>(v);
    case 1: return quad_perm_i<0x55>(v);
    case 2: return quad_perm_i<0xAA>(v);
    default: return quad_perm_i<0xFF>(v);
  }
}
__device__ __forceinline__ float quad_bcast_f(float v, int k) {
  return __builtin_bit_cast(float, quad_bcast_i(__builtin_bit_cast(int, v), k));
}
// sum over the 4 lanes of a quad: lane ^ 1 (quad_perm [1,0,3,2]) then lane ^ 2 ([2,3,0,1])
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, quad_perm_i<0xB1>(__builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, quad_perm_i<0x4E>(__builtin_bit_cast(int, v)));
  return v;
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
  const float* lg = logits + bq * dm.lg_rs + m * np;
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
    const BoxGeo g = make_box(ref + bq * 7, off + bq * dm.off_rs + (m * dm.l + li) * dm.v, dm.v);
    if (dm.lp >= 4 && (long long)dm.s * row_stride < (1ll << 31)) {   // (32-bit row offsets)
      // The lanes of a pair differ only in their channels, and the first version had every one of them work out
      // every point's geometry (110 VALU instructions per point, 50 of them geometry, in a kernel whose 100 gathers
      // per lane leave the vector-memory pipe half idle).  Now each lane of a QUAD does the geometry of one point in
      // four and hands the 4 row offsets, the 4 corner weights and the attention weight to the quad by DPP; the
      // arithmetic on the values is unchanged, bit for bit.
      const int qc = lane & 3;
      for (int p0 = 0; p0 < dm.p; p0 += 4) {
        const int op = p0 + qc;
        const bool own = op < dm.p;
        const BoxPx px = box_point(g, own ? kidx[op * 2] : 0.f, own ? kidx[op * 2 + 1] : 0.f, H, W);
        const float wgt_o = own ? as[li * dm.p + op] * inv : 0.f;
        const float h_im = px.h_im, w_im = px.w_im;
        const int ins_o = own && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lwf;
        const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
        const int y0 = min(max(h_low, 0), H - 1), y1 = max(min(h_low + 1, H - 1), 0);
        const int x0 = min(max(w_low, 0), W - 1), x1 = max(min(w_low + 1, W - 1), 0);
        const int o1 = (y0 * W + x0) * row_stride, o2 = (y0 * W + x1) * row_stride;
        const int o3 = (y1 * W + x0) * row_stride, o4 = (y1 * W + x1) * row_stride;
        const float w1_o = (t_ok && l_ok) ? hh * hw : 0.f, w2_o = (t_ok && r_ok) ? hh * lwf : 0.f;
        const float w3_o = (b_ok && l_ok) ? lh * hw : 0.f, w4_o = (b_ok && r_ok) ? lh * lwf : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ins = quad_bcast_i(ins_o, c);
          const int a1 = quad_bcast_i(o1, c), a2 = quad_bcast_i(o2, c), a3 = quad_bcast_i(o3, c), a4 = quad_bcast_i(o4, c);
          const float w1 = quad_bcast_f(w1_o, c), w2 = quad_bcast_f(w2_o, c), w3 = quad_bcast_f(w3_o, c),
                      w4 = quad_bcast_f(w4_o, c), wgt = quad_bcast_f(wgt_o, c);
          if (ins && active) {
            const float4 v1 = ld4(v + a1), v2 = ld4(v + a2), v3 = ld4(v + a3), v4 = ld4(v + a4);
            acc.x = fmaf(bil(w1, w2, w3, w4, v1.x, v2.x, v3.x, v4.x), wgt, acc.x);
            acc.y = fmaf(bil(w1, w2, w3, w4, v1.y, v2.y, v3.y, v4.y), wgt, acc.y);
            acc.z = fmaf(bil(w1, w2, w3, w4, v1.z, v2.z, v3.z, v4.z), wgt, acc.z);
            acc.w = fmaf(bil(w1, w2, w3, w4, v1.w, v2.w, v3.w, v4.w), wgt, acc.w);
          }
        }
      }
      continue;
    }
    for (int pi = 0; pi < dm.p; ++pi) {
      const BoxPx px = box_point(g, kidx[pi * 2], kidx[pi * 2 + 1], H, W);
      const float wgt = as[li * dm.p + pi] * inv;
      const float h_im = px.h_im, w_im = px.w_im;
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
               float* __restrict__ grad_logits, int* __restrict__ cursor, int2* __restrict__ entries,
               const int* __restrict__ bin_end, int* __restrict__ overflow) {
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
    const float* lg = logits + bq * dm.lg_rs + m * np;
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
      const BoxGeo g = make_box(ref + bq * 7, off + bq * dm.off_rs + (m * dm.l + li) * dm.v, dm.v);
      float dcx = 0.f, dcy = 0.f, dw = 0.f, dh = 0.f, dth = 0.f;
      for (int pi = 0; pi < dm.p; ++pi) {
        const float kxn = kidx[pi * 2], kyn = kidx[pi * 2 + 1];
        const BoxPx px = box_point(g, kxn, kyn, H, W);
        const float gx = px.gx, gy = px.gy;
        const float wgt = as[li * dm.p + pi] * inv;
        const float h_im = px.h_im, w_im = px.w_im;
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
              } else if (cursor) {
                // binned path: one (row of grad_out, weight) entry per corner instead of D float atomics; the
                // entries are summed per (cell, head) by box_bin_reduce_kernel
                if (sub == 0) {
                  const long long bin = ((long long)bi * dm.s + starts[li] + (long long)cy * W + cx) * dm.h + m;
                  bin_push(cursor, bin_end, overflow, entries, bin, (int)t, wc[cn] * wgt);
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
        float* go = grad_off + bq * dm.off_rs + (m * dm.l + li) * dm.v;
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
      for (int e = sub; e < np; e += LP) grad_logits[bq * dm.lg_rs + m * np + e] = as[e] * inv * (ga_s[slot][e] - dot);
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

// ---- encoder backward in GEMM form ---------------------------------------------------------------
// Encoder self-attention: the queries ARE the cells of the (single-level) value map, so an 8x8 query tile only
// touches the 16x16 window of cells around it.  Per (tile, head, scene) the whole backward then factors into two
// small dense products plus SCALAR scatter / gather work -- no per-channel atomics and no per-point channel
// gathers (the previous kernel issued 3200 fp64 LDS atomics and 100 float4 gathers per (query, head)):
//
//   G [64 q  x 256 cells] = GO [64 x 32] . V^T [32 x 256]        (MFMA)   "what each cell's value says to q"
//       grad_attn(q, p)   = sum_corners wc * G[q][cell]                    4 scalar LDS reads per point
//       grad_loc (q, p)   = bilinear differences of the same 4 scalars
//   W [256 cells x 64 q]  += attn * wc   at the 4 corner cells             plain LDS read-modify-write: column q
//                                                                          is private to its 4 corner lanes
//   GV[256 cells x 32]    = W [256 x 64] . GO [64 x 32]           (MFMA)   -> one global atomic per touched
//                                                                          (cell, channel) of the window
//
// G and W share one LDS buffer (pass A reads G, pass B rebuilds the geometry and fills W).  Sampling points
// that leave the window (boxes grown past R = 4 cells) take a per-channel global path, as before.
// Tile geometry of box_bwd_tile_kernel: TQY x 8 queries of one head, the window of R cells around them.
//   TQY = 8: 64 queries, 16 x 16 window, 512 threads, 132 KB of LDS -> ONE workgroup per CU, all of its waves in the same
//            phase (the round-2 kernel);
//   TQY = 4: 32 queries, 12 x 16 window, 256 threads,  68 KB of LDS -> TWO workgroups per CU, whose phases (MFMA / LDS
//            scatter / global flush) overlap; 1.5x the halo cells per query.
template <int TQY_>
struct BT {
  static constexpr int TQY = TQY_, TQX = 8, R = 4, WINY = TQY + 2 * R, WINX = TQX + 2 * R, NQ = TQY * TQX, NC = WINY * WINX, D = 32;
  static constexpr int VS = 36;        // row stride of V / GO tiles (floats, 16-byte aligned rows)
  static constexpr int GS = NC + 4;    // G[q][cell]
  static constexpr int WS = NQ + 4;    // W[cell][q]
  static constexpr int kThreads = NQ * 8;
  static constexpr int kGW = (NQ * GS > NC * WS) ? NQ * GS : NC * WS;
  static constexpr int NW = kThreads / 64, MB = NQ / 16, NB = NC / 16;   // waves, 16-query blocks, 16-cell blocks
  static constexpr int WPM = NW / MB, NBW = NB / WPM, CBW = NB / NW;     // S1: waves per query block, cell blocks per wave; S5
  static_assert(MB * WPM == NW && WPM * NBW == NB && NW * CBW == NB && NBW % 2 == 0, "tile shape does not divide over the waves");
};
namespace bt {
constexpr int PMAX = 32;  // L * P of the fused encoder path
constexpr int TQX = 8, R = 4;
template <int TQY>
constexpr size_t lds_bytes() {
  using B = BT<TQY>;
  return sizeof(float) * (B::NC * B::VS + B::NQ * B::VS + B::kGW + 2 * B::NQ * PMAX + 2 * PMAX);
}
}  // namespace bt

template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, true));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TQY>
__global__ void __launch_bounds__(BT<TQY>::kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
box_bwd_tile_kernel(const float* __restrict__ value, const long long* __restrict__ shapes,
                    const float* __restrict__ ref, const float* __restrict__ off, const float* __restrict__ logits,
                    const float* __restrict__ kidx, const float* __restrict__ grad_out, BoxDims dm,
                    float* __restrict__ grad_value, float* __restrict__ grad_off, float* __restrict__ grad_logits,
                    int* __restrict__ cursor, int2* __restrict__ entries, const int* __restrict__ bin_end,
                    int* __restrict__ overflow, int color) {
  using B = BT<TQY>;
  constexpr int TQX = B::TQX, R = B::R, WINY = B::WINY, WINX = B::WINX, NQ = B::NQ, NC = B::NC, D = B::D, VS = B::VS, GS = B::GS,
                WS = B::WS, kThreads = B::kThreads, kGW = B::kGW, PMAX = bt::PMAX;
  extern __shared__ float lds[];
  float* Vs = lds;                     // [NC][VS]
  float* GOs = Vs + NC * VS;           // [NQ][VS]
  float* GW = GOs + NQ * VS;           // G [NQ][GS]  |  W [NC][WS]
  float* a_s = GW + kGW;               // [NQ][PMAX] un-normalised softmax weights
  float* ga_s = a_s + NQ * PMAX;       // [NQ][PMAX]
  float* k_s = ga_s + NQ * PMAX;       // [PMAX][2] the k x k lattice (read at every point: keep it out of global)
  const int Hm = (int)shapes[0], Wm = (int)shapes[1];
  if ((long long)Hm * Wm != dm.s) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 2 * dm.p) k_s[tid] = kidx[tid];
  const int slot = tid >> 3, sub = tid & 7, corner = sub & 3, half = sub >> 2;
  const int m = blockIdx.y, bi = blockIdx.z;
  const int np = dm.p;  // single level
  const int tiles_x = (Wm + TQX - 1) / TQX;
  const int tiles_y = (Hm + TQY - 1) / TQY;
  // color >= 0: this launch works on the tiles (ty, tx) with (ty % NCY, tx % NCX) == (color / NCX, color % NCX) only.  The
  // windows of two such tiles never overlap (a window is the tile + R cells all around: NCY x NCX tiles), so the window
  // flush is a plain read-modify-write and the NCY * NCX launches apply the overlapping windows' contributions to a cell
  // in a FIXED order: grad_value is reproducible bit for bit.  color < 0: one launch over all tiles, float atomics.
  constexpr int NCY = (WINY + TQY - 1) / TQY, NCX = (WINX + TQX - 1) / TQX;
  const int cy0 = color >= 0 ? color / NCX : 0, cx0 = color >= 0 ? color % NCX : 0;
  const int sy = color >= 0 ? NCY : 1, sx = color >= 0 ? NCX : 1;
  const int ctx = (tiles_x - cx0 + sx - 1) / sx, cty = (tiles_y - cy0 + sy - 1) / sy;   // tiles of this launch
  const int ntiles = max(ctx, 0) * max(cty, 0);
  auto tile_of = [&](int j) { return (cy0 + sy * (j / ctx)) * tiles_x + cx0 + sx * (j % ctx); };
  const long long S = dm.s;
  constexpr int VPT = NC * (D / 4) / kThreads;  // float4 of the value window per thread (4)
  constexpr int EPT = PMAX / 2;                 // points per lane (one half of the lattice)

  // Everything a work item reads from global memory, fetched one item ahead: with 132 KB of LDS there is one
  // workgroup per CU and all of its waves sit in the same phase, so nothing else would hide the latency.
  struct Pre {
    float4 v[VPT];
    float4 go;
    float lg[4], rf[5], of[5];
  } pre;
  auto fetch = [&](int j) {
    const int tile = tile_of(j);
    const int ty0 = (tile / tiles_x) * TQY, tx0 = (tile % tiles_x) * TQX;
#pragma unroll
    for (int it = 0; it < VPT; ++it) {
      const int idx = tid + it * kThreads;
      const int cell = idx >> 3, c4 = (idx & 7) * 4;
      const int cy = min(max(ty0 - R + cell / WINX, 0), Hm - 1), cx = min(max(tx0 - R + cell % WINX, 0), Wm - 1);
      pre.v[it] = ld4(value + (((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m) * D + c4);
    }
    const int qy = min(ty0 + slot / TQX, Hm - 1), qx = min(tx0 + slot % TQX, Wm - 1);
    const long long bq = (long long)bi * dm.lq + (long long)qy * Wm + qx, t = bq * dm.h + m;
    pre.go = ld4(grad_out + t * D + sub * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) pre.lg[k] = logits[bq * dm.lg_rs + m * np + min(sub + 8 * k, np - 1)];
    const float* r = ref + bq * 7;
    pre.rf[0] = r[0];
    pre.rf[1] = r[1];
    pre.rf[2] = r[3];
    pre.rf[3] = r[4];
    pre.rf[4] = r[6];
#pragma unroll
    for (int k = 0; k < 5; ++k) pre.of[k] = off[bq * dm.off_rs + m * dm.v + min(k, dm.v - 1)];
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);

  for (int tj = blockIdx.x; tj < ntiles; tj += gridDim.x) {
    const int tile = tile_of(tj);
    const int ty0 = (tile / tiles_x) * TQY, tx0 = (tile % tiles_x) * TQX;
    const int wy0 = ty0 - R, wx0 = tx0 - R;
    __syncthreads();  // previous item's GEMM-2 is done with GOs / W
    // ---- S0: registers -> LDS ------------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < VPT; ++it) {
      const int idx = tid + it * kThreads;
      const int cell = idx >> 3, c4 = (idx & 7) * 4;
      const int cy = wy0 + cell / WINX, cx = wx0 + cell % WINX;
      const bool in = cy >= 0 && cy < Hm && cx >= 0 && cx < Wm;
      *reinterpret_cast<float4*>(Vs + cell * VS + c4) = in ? pre.v[it] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int qy = ty0 + slot / TQX, qx = tx0 + slot % TQX;
    const bool qok = qy < Hm && qx < Wm;
    const long long bqs = qok ? ((long long)bi * dm.lq + (long long)qy * Wm + qx) : 0;
    const long long t = bqs * dm.h + m;
    *reinterpret_cast<float4*>(GOs + slot * VS + sub * 4) = qok ? pre.go : make_float4(0.f, 0.f, 0.f, 0.f);
    // softmax statistics of the pair (8 lanes share the work)
    float* as = a_s + slot * PMAX;
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sub + 8 * k < np) mx = fmaxf(mx, pre.lg[k]);
#pragma unroll
    for (int dlt = 4; dlt > 0; dlt >>= 1) mx = fmaxf(mx, __shfl_xor(mx, dlt, 64));
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sub + 8 * k < np) {
        const float ex = expf(pre.lg[k] - mx);
        as[sub + 8 * k] = ex;
        den += ex;
      }
#pragma unroll
    for (int dlt = 4; dlt > 0; dlt >>= 1) den += __shfl_xor(den, dlt, 64);
    const float inv = 1.0f / den;
    const BoxGeo g = make_box_v(pre.rf, pre.of, dm.v);
    __syncthreads();
    if (tj + (int)gridDim.x < ntiles) fetch(tj + gridDim.x);  // in flight during the phases below

    // ---- S1: G = GO . V^T ----------------------------------------------------------------------------------
    {
      const int mb = wave / B::WPM, nb0 = (wave % B::WPM) * B::NBW;
      const int r16 = lane & 15, kk = lane >> 4;
      float a[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) a[ks] = GOs[(16 * mb + r16) * VS + 4 * ks + kk];
#pragma unroll
      for (int nbi = 0; nbi < B::NBW; nbi += 2) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* b0 = Vs + (16 * (nb0 + nbi) + r16) * VS + kk;
        const float* b1 = b0 + 16 * VS;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b0[4 * ks], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b1[4 * ks], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* gp = GW + (16 * mb + 4 * kk + r) * GS + 16 * (nb0 + nbi) + r16;
          gp[0] = acc0[r];
          gp[16] = acc1[r];
        }
      }
    }
    __syncthreads();

    // ---- S2 (pass A): this lane = (query, corner, half of the lattice).  Gradients wrt the attention weights
    // and the sampling locations come from 4 scalars of G per point; the (cell, weight) of the lane's corner is
    // kept in registers for pass B.
    // The kernel is bound by its VALU instruction count (static count of this pass in the first version: 2400 of the
    // 5300 per tile and wave), and there all 4 corner lanes of a point redid the point's geometry and its gradient
    // arithmetic.  Now the quad works on 4 points at a time: each lane does the geometry of ITS point of the group,
    // then for each of the 4 points the quad takes cell / fractions / weight from the owner lane by DPP, every lane reads
    // G at its corner (the LDS access pattern that the quad mapping keeps conflict-free) and the owner collects the 4
    // corner values; the gradient arithmetic runs once per point, on the owner.
    float dcx = 0.f, dcy = 0.f, dw = 0.f, dh = 0.f, dth = 0.f, dot = 0.f;
    int e_cell[EPT];    // window cell (>= 0), -1: nothing to add, -2: outside the window (global path)
    float e_w[EPT];
    const int cyo = corner >> 1, cxo = corner & 1;
    const float sy = cyo ? 1.f : -1.f, ay = cyo ? 0.f : 1.f, sx = cxo ? 1.f : -1.f, ax = cxo ? 0.f : 1.f;
#pragma unroll
    for (int j = 0; j < EPT / 4; ++j) {
      const int op = half + 2 * (4 * j + corner);   // the point this lane owns in group j
      const bool own = qok && op < np;
      const float kxn = own ? k_s[op * 2] : 0.f, kyn = own ? k_s[op * 2 + 1] : 0.f;
      const BoxPx px = box_point(g, kxn, kyn, Hm, Wm);
      const float wgt = own ? as[op] * inv : 0.f;
      const float h_im = px.h_im, w_im = px.w_im;
      const bool inside = own && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hm) && (w_im < (float)Wm);
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
      const int h_pub = inside ? h_low : -(1 << 20);   // no corner of a point that samples nothing is "ok"
      float gq[4] = {0.f, 0.f, 0.f, 0.f};              // G at the 4 corners of the OWN point
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 4 * j + c;
        const int hb = quad_bcast_i(h_pub, c), wb = quad_bcast_i(w_low, c);
        const float lhb = quad_bcast_f(lh, c), lwb = quad_bcast_f(lwf, c), wgb = quad_bcast_f(wgt, c);
        const int cy = hb + cyo, cx = wb + cxo;
        const bool ok = (unsigned)cy < (unsigned)Hm && (unsigned)cx < (unsigned)Wm;
        const int ly = cy - wy0, lx = cx - wx0;
        const bool in_win = (unsigned)ly < (unsigned)WINY && (unsigned)lx < (unsigned)WINX;
        e_cell[k] = -1;
        e_w[k] = 0.f;
        float gval = 0.f;
        if (ok) {
          // ((corner >> 1) ? lh : 1 - lh) * ((corner & 1) ? lw : 1 - lw): one rounding each, as the select form
          e_w[k] = wgb * (fmaf(sy, lhb, ay) * fmaf(sx, lwb, ax));
          if (in_win) {
            e_cell[k] = ly * WINX + lx;
            gval = GW[slot * GS + e_cell[k]];
          } else {  // the box has grown out of the window: dot product against the global value row
            e_cell[k] = -2;
            // (rare on a fresh model, common once the boxes have grown: 16-byte loads, the same 32 additions in order)
            const float4* vr = reinterpret_cast<const float4*>(value + (((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m) * D);
            const float4* gr = reinterpret_cast<const float4*>(GOs + slot * VS);
#pragma unroll 1
            for (int cc = 0; cc < D / 4; ++cc) {
              const float4 v4 = vr[cc], g4 = gr[cc];
              gval = fmaf(g4.w, v4.w, fmaf(g4.z, v4.z, fmaf(g4.y, v4.y, fmaf(g4.x, v4.x, gval))));
            }
          }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float t4 = quad_bcast_f(gval, cc);
          gq[cc] = (corner == c) ? t4 : gq[cc];
        }
      }
      const float g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3];
      const float hh = 1.f - lh, hw = 1.f - lwf;
      const float gx = px.gx, gy = px.gy;
      const float ga = fmaf(lh * lwf, g3, fmaf(lh * hw, g2, fmaf(hh * lwf, g1, hh * hw * g0)));
      const float gwl = (float)Wm * wgt * fmaf(hh, g1 - g0, lh * (g3 - g2));
      const float ghl = (float)Hm * wgt * fmaf(hw, g2 - g0, lwf * (g3 - g1));
      dcx += gwl;
      dcy += ghl;
      dw += kxn * (gwl * g.cs + ghl * g.sn);
      dh += kyn * (ghl * g.cs - gwl * g.sn);
      dth += gwl * (-(gx * g.sn) - gy * g.cs) + ghl * (gx * g.cs - gy * g.sn);
      dot = fmaf(wgt, ga, dot);
      if (own) ga_s[slot * PMAX + op] = ga;
    }
    // every lane holds the sums over its own points: 4 corner lanes of a half, then the two halves 4 lanes apart
    dcx = quad_sum(dcx);
    dcy = quad_sum(dcy);
    dw = quad_sum(dw);
    dh = quad_sum(dh);
    dth = quad_sum(dth);
    dot = quad_sum(dot);
    dcx += __shfl_xor(dcx, 4, 64);
    dcy += __shfl_xor(dcy, 4, 64);
    dw += __shfl_xor(dw, 4, 64);
    dh += __shfl_xor(dh, 4, 64);
    dth += __shfl_xor(dth, 4, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (qok && sub == 0) {
      float* go = grad_off + bqs * dm.off_rs + m * dm.v;
      go[0] = dcx * g.rw / 8.0f;
      go[1] = dcy * g.rh / 8.0f;
      go[2] = g.w_on ? dw * g.rw / 8.0f : 0.0f;
      go[3] = g.h_on ? dh * g.rh / 8.0f : 0.0f;
      if (dm.v == 5) go[4] = dth * (2.0f * 3.14159274f / 16.0f);
    }
    __syncthreads();  // G fully consumed; ga_s complete
    if (qok)
      for (int e = sub; e < np; e += 8) grad_logits[bqs * dm.lg_rs + m * np + e] = as[e] * inv * (ga_s[slot * PMAX + e] - dot);

    // ---- S3 / S4 (pass B): W[cell][q] = sum of attn * bilinear weight ---------------------------------------
    for (int i = tid; i < NC * WS / 4; i += kThreads) reinterpret_cast<float4*>(GW)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // Column `slot` of W belongs to the 8 lanes of the query.  At one step the 4 corner lanes of a half hit 4
    // distinct cells, so a plain read-modify-write is safe; the two halves take turns (their points may share
    // a cell), and the LDS pipe keeps consecutive steps of a wave in order.  (LDS float atomics instead of the 32
    // dependent round trips: 490 -> 726 us per launch, ds_add_f32 is far slower than the read + write it replaces.)
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        if (half == hsel && e_cell[k] >= 0) GW[e_cell[k] * WS + slot] += e_w[k];
        asm volatile("" ::: "memory");
      }
    }
    // corners outside the window.  As the boxes grow with training these stop being rare, and 32 float atomics per corner
    // are a cliff (2 x 35 344 queries with boxes 3x their anchors: 0.67 -> 39.8 ms per launch).  With the binned path
    // (`cursor`: the launch code counted these corners per (cell, head) row and scanned the counts) a corner is ONE
    // 8-byte entry (grad_out row, weight) that box_bin_reduce_kernel sums per row after this kernel.
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (e_cell[k] == -2) {
        const int pi = half + 2 * k;
        const BoxPx px = box_point(g, k_s[pi * 2], k_s[pi * 2 + 1], Hm, Wm);
        const int cy = (int)floorf(px.h_im) + (corner >> 1), cx = (int)floorf(px.w_im) + (corner & 1);
        const long long bin = ((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m;
        if (cursor) {
          bin_push(cursor, bin_end, overflow, entries, bin, (int)t, e_w[k]);
        } else {
          float* gv = grad_value + bin * D;
#pragma unroll 1
          for (int c = 0; c < D; ++c) unsafeAtomicAdd(gv + c, e_w[k] * GOs[slot * VS + c]);
        }
      }
    }
    __syncthreads();

    // ---- S5: GV = W . GO, flushed with one atomic per touched (cell, channel) -------------------------------
    {
      const int r16 = lane & 15, kk = lane >> 4;
      constexpr int CBW = B::CBW;   // 16-cell blocks of W per wave
      f32x4 acc[CBW][2];
#pragma unroll
      for (int i = 0; i < CBW; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      // coloured launch: the window's current contents, requested BEFORE the product so that the read half of the
      // read-modify-write flush is hidden behind the MFMAs
      float cur[CBW][2][4];
      if (color >= 0) {
#pragma unroll
        for (int i = 0; i < CBW; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cell = 16 * (CBW * wave + i) + 4 * kk + r;
            const int cy = min(max(wy0 + cell / WINX, 0), Hm - 1), cx = min(max(wx0 + cell % WINX, 0), Wm - 1);
            const float* gv = grad_value + (((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m) * D + r16;
            cur[i][0][r] = gv[0];
            cur[i][1][r] = gv[16];
          }
      }
      const float* ap = GW + (16 * (CBW * wave) + r16) * WS + kk;
      const float* bp = GOs + kk * VS + r16;
#pragma unroll
      for (int ks = 0; ks < NQ / 4; ++ks) {
        const float b0 = bp[4 * ks * VS], b1 = bp[4 * ks * VS + 16];
#pragma unroll
        for (int i = 0; i < CBW; ++i) {
          const float av = ap[i * 16 * WS + 4 * ks];
          acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[i][1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cell = 16 * (CBW * wave + i) + 4 * kk + r;
          const int cy = wy0 + cell / WINX, cx = wx0 + cell % WINX;
          if (cy >= 0 && cy < Hm && cx >= 0 && cx < Wm) {
            float* gv = grad_value + (((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m) * D + r16;
            if (color >= 0) {   // nobody else touches this window during this launch
              if (acc[i][0][r] != 0.0f) gv[0] = __fadd_rn(cur[i][0][r], acc[i][0][r]);
              if (acc[i][1][r] != 0.0f) gv[16] = __fadd_rn(cur[i][1][r], acc[i][1][r]);
            } else {
              if (acc[i][0][r] != 0.0f) unsafeAtomicAdd(gv, acc[i][0][r]);
              if (acc[i][1][r] != 0.0f) unsafeAtomicAdd(gv + 16, acc[i][1][r]);
            }
          }
        }
    }
  }  // tile loop
}

// ---- the same backward as TWO kernels (round 6) -------------------------------------------------------------------------------
// box_bwd_tile_kernel holds, per 4 x 8-query tile, the value window, the grad_out tile, G / W, two softmax tables AND the
// (cell, weight) pairs of pass A for pass B in registers: 256 VGPRs + 68 KB of LDS = two waves per SIMD, which issue a third
// of their life (SQ counters, profiles/r05c_pmc_box_bwd_tile_*.txt): latency-bound, and three re-mappings of the same tile
// did not change that.  Split along its only internal boundary, each half fits FOUR waves per SIMD:
//   box_bwd_tile_a_kernel  S0 + S1 + pass A: G = GO . V^T, gradients of the offsets and logits.  G is written over the value
//                          window (every wave holds its G blocks in registers until all reads of V are done): 40 KB of LDS.
//                          No colour classes -- it writes nothing that overlaps -- so ONE launch over all tiles.
//   box_bwd_tile_b_kernel  softmax + geometry again (a few hundred VALU instructions per tile against two fewer waves), the
//                          W scatter in the order of the one-kernel form, GV = W . GO, window flush: 37 KB of LDS; per
//                          colour class (plain read-modify-write) or one launch with atomics, as before.
// Same arithmetic in the same order per element: grad_value / grad_offsets / grad_logits are the one-kernel form's bit for bit
// (tests/test_box_fused_gpu.py::test_split_backward_bits_equal_one_kernel).  EFG_BOX_SPLIT=0: the one-kernel form (A/B).
namespace bt {
template <int TQY>
constexpr size_t lds_bytes_a() {
  using B = BT<TQY>;
  constexpr int kVG = (B::NC * B::VS > B::NQ * B::GS) ? B::NC * B::VS : B::NQ * B::GS;
  return sizeof(float) * (kVG + B::NQ * B::VS + 2 * B::NQ * PMAX + 2 * PMAX);
}
template <int TQY>
constexpr size_t lds_bytes_b() {
  using B = BT<TQY>;
  return sizeof(float) * (B::NC * B::WS + B::NQ * B::VS + B::NQ * PMAX + 2 * PMAX);
}
}  // namespace bt

template <int TQY>
__global__ void __launch_bounds__(BT<TQY>::kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
box_bwd_tile_a_kernel(const float* __restrict__ value, const long long* __restrict__ shapes, const float* __restrict__ ref,
                      const float* __restrict__ off, const float* __restrict__ logits, const float* __restrict__ kidx,
                      const float* __restrict__ grad_out, BoxDims dm, float* __restrict__ grad_off,
                      float* __restrict__ grad_logits, int* __restrict__ counts, unsigned* __restrict__ absmax) {
  // counts / absmax (both or neither): what box_bin_count_kernel and absmax_bits_kernel compute for the binned far corners,
  // taken along -- this pass classifies every corner with the test pass B pushes with and reads every grad_out row once
  using B = BT<TQY>;
  constexpr int TQX = B::TQX, R = B::R, WINY = B::WINY, WINX = B::WINX, NQ = B::NQ, NC = B::NC, D = B::D, VS = B::VS, GS = B::GS,
                kThreads = B::kThreads, PMAX = bt::PMAX;
  unsigned go_max = 0u;
  constexpr int kVG = (NC * VS > NQ * GS) ? NC * VS : NQ * GS;
  extern __shared__ float lds[];
  float* VG = lds;                     // V [NC][VS] in S0 / S1, then G [NQ][GS]
  float* GOs = VG + kVG;               // [NQ][VS]
  float* a_s = GOs + NQ * VS;          // [NQ][PMAX] un-normalised softmax weights
  float* ga_s = a_s + NQ * PMAX;       // [NQ][PMAX]
  float* k_s = ga_s + NQ * PMAX;       // [PMAX][2]
  const int Hm = (int)shapes[0], Wm = (int)shapes[1];
  if ((long long)Hm * Wm != dm.s) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 2 * dm.p) k_s[tid] = kidx[tid];
  const int slot = tid >> 3, sub = tid & 7, corner = sub & 3, half = sub >> 2;
  const int m = blockIdx.y, bi = blockIdx.z;
  const int np = dm.p;
  const int tiles_x = (Wm + TQX - 1) / TQX, tiles_y = (Hm + TQY - 1) / TQY;
  const int ntiles = tiles_x * tiles_y;
  const long long S = dm.s;
  constexpr int VPT = NC * (D / 4) / kThreads;
  constexpr int EPT = PMAX / 2;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ty0 = (tile / tiles_x) * TQY, tx0 = (tile % tiles_x) * TQX;
    const int wy0 = ty0 - R, wx0 = tx0 - R;
    __syncthreads();  // the previous tile is done with the LDS
    // ---- S0: global -> LDS (four workgroups per CU: the others' phases hide these loads) -------------------------
    {
      float4 v[VPT];
#pragma unroll
      for (int it = 0; it < VPT; ++it) {
        const int idx = tid + it * kThreads;
        const int cell = idx >> 3, c4 = (idx & 7) * 4;
        const int cy = min(max(wy0 + cell / WINX, 0), Hm - 1), cx = min(max(wx0 + cell % WINX, 0), Wm - 1);
        v[it] = ld4(value + (((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m) * D + c4);
      }
#pragma unroll
      for (int it = 0; it < VPT; ++it) {
        const int idx = tid + it * kThreads;
        const int cell = idx >> 3, c4 = (idx & 7) * 4;
        const int cy = wy0 + cell / WINX, cx = wx0 + cell % WINX;
        const bool in = cy >= 0 && cy < Hm && cx >= 0 && cx < Wm;
        *reinterpret_cast<float4*>(VG + cell * VS + c4) = in ? v[it] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const int qy = ty0 + slot / TQX, qx = tx0 + slot % TQX;
    const bool qok = qy < Hm && qx < Wm;
    const long long bqs = qok ? ((long long)bi * dm.lq + (long long)qy * Wm + qx) : 0;
    const long long t = bqs * dm.h + m;
    {
      const int qyc = min(qy, Hm - 1), qxc = min(qx, Wm - 1);
      const long long bqc = (long long)bi * dm.lq + (long long)qyc * Wm + qxc;
      const float4 go = ld4(grad_out + (bqc * dm.h + m) * D + sub * 4);
      *reinterpret_cast<float4*>(GOs + slot * VS + sub * 4) = qok ? go : make_float4(0.f, 0.f, 0.f, 0.f);
      if (qok) {
        go_max = max(max(go_max, __float_as_uint(go.x) & 0x7fffffffu), __float_as_uint(go.y) & 0x7fffffffu);
        go_max = max(max(go_max, __float_as_uint(go.z) & 0x7fffffffu), __float_as_uint(go.w) & 0x7fffffffu);
      }
    }
    float lgv[4], rf[5], of[5];
    {
      const int qyc = min(qy, Hm - 1), qxc = min(qx, Wm - 1);
      const long long bqc = (long long)bi * dm.lq + (long long)qyc * Wm + qxc;
#pragma unroll
      for (int k = 0; k < 4; ++k) lgv[k] = logits[bqc * dm.lg_rs + m * np + min(sub + 8 * k, np - 1)];
      const float* r = ref + bqc * 7;
      rf[0] = r[0], rf[1] = r[1], rf[2] = r[3], rf[3] = r[4], rf[4] = r[6];
#pragma unroll
      for (int k = 0; k < 5; ++k) of[k] = off[bqc * dm.off_rs + m * dm.v + min(k, dm.v - 1)];
    }
    float* as = a_s + slot * PMAX;
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sub + 8 * k < np) mx = fmaxf(mx, lgv[k]);
#pragma unroll
    for (int dlt = 4; dlt > 0; dlt >>= 1) mx = fmaxf(mx, __shfl_xor(mx, dlt, 64));
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sub + 8 * k < np) {
        const float ex = expf(lgv[k] - mx);
        as[sub + 8 * k] = ex;
        den += ex;
      }
#pragma unroll
    for (int dlt = 4; dlt > 0; dlt >>= 1) den += __shfl_xor(den, dlt, 64);
    const float inv = 1.0f / den;
    const BoxGeo g = make_box_v(rf, of, dm.v);
    __syncthreads();

    // ---- S1: G = GO . V^T, the wave's blocks kept in registers until every wave has read its part of V --------------
    {
      const int mb = wave / B::WPM, nb0 = (wave % B::WPM) * B::NBW;
      const int r16 = lane & 15, kk = lane >> 4;
      float a[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) a[ks] = GOs[(16 * mb + r16) * VS + 4 * ks + kk];
      f32x4 acc[B::NBW];
#pragma unroll
      for (int nbi = 0; nbi < B::NBW; nbi += 2) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* b0 = VG + (16 * (nb0 + nbi) + r16) * VS + kk;
        const float* b1 = b0 + 16 * VS;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b0[4 * ks], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b1[4 * ks], acc1, 0, 0, 0);
        }
        acc[nbi] = acc0;
        acc[nbi + 1] = acc1;
      }
      __syncthreads();   // V is dead: G goes over it
#pragma unroll
      for (int nbi = 0; nbi < B::NBW; ++nbi)
#pragma unroll
        for (int r = 0; r < 4; ++r) VG[(16 * mb + 4 * kk + r) * GS + 16 * (nb0 + nbi) + r16] = acc[nbi][r];
    }
    __syncthreads();

    // ---- S2 (pass A), as in box_bwd_tile_kernel -----------------------------------------------------------------
    float dcx = 0.f, dcy = 0.f, dw = 0.f, dh = 0.f, dth = 0.f, dot = 0.f;
    const int cyo = corner >> 1, cxo = corner & 1;
#pragma unroll 1
    for (int j = 0; j < EPT / 4; ++j) {
      const int op = half + 2 * (4 * j + corner);
      const bool own = qok && op < np;
      const float kxn = own ? k_s[op * 2] : 0.f, kyn = own ? k_s[op * 2 + 1] : 0.f;
      const BoxPx px = box_point(g, kxn, kyn, Hm, Wm);
      const float wgt = own ? as[op] * inv : 0.f;
      const float h_im = px.h_im, w_im = px.w_im;
      const bool inside = own && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hm) && (w_im < (float)Wm);
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
      const int h_pub = inside ? h_low : -(1 << 20);
      float gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int hb = quad_bcast_i(h_pub, c), wb = quad_bcast_i(w_low, c);
        const int cy = hb + cyo, cx = wb + cxo;
        const bool ok = (unsigned)cy < (unsigned)Hm && (unsigned)cx < (unsigned)Wm;
        const int ly = cy - wy0, lx = cx - wx0;
        const bool in_win = (unsigned)ly < (unsigned)WINY && (unsigned)lx < (unsigned)WINX;
        float gval = 0.f;
        if (ok) {
          if (in_win) {
            gval = VG[slot * GS + ly * WINX + lx];
          } else {  // the box has grown out of the window: dot product against the global value row
            const long long cellrow = ((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m;
            if (counts) atomicAdd(counts + cellrow, 1);   // one entry of pass B in the bin of this (cell, head) row
            const float4* vr = reinterpret_cast<const float4*>(value + cellrow * D);
            const float4* gr = reinterpret_cast<const float4*>(GOs + slot * VS);
#pragma unroll 1
            for (int cc = 0; cc < D / 4; ++cc) {
              const float4 v4 = vr[cc], g4 = gr[cc];
              gval = fmaf(g4.w, v4.w, fmaf(g4.z, v4.z, fmaf(g4.y, v4.y, fmaf(g4.x, v4.x, gval))));
            }
          }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float t4 = quad_bcast_f(gval, cc);
          gq[cc] = (corner == c) ? t4 : gq[cc];
        }
      }
      const float g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3];
      const float hh = 1.f - lh, hw = 1.f - lwf;
      const float gx = px.gx, gy = px.gy;
      const float ga = fmaf(lh * lwf, g3, fmaf(lh * hw, g2, fmaf(hh * lwf, g1, hh * hw * g0)));
      const float gwl = (float)Wm * wgt * fmaf(hh, g1 - g0, lh * (g3 - g2));
      const float ghl = (float)Hm * wgt * fmaf(hw, g2 - g0, lwf * (g3 - g1));
      dcx += gwl;
      dcy += ghl;
      dw += kxn * (gwl * g.cs + ghl * g.sn);
      dh += kyn * (ghl * g.cs - gwl * g.sn);
      dth += gwl * (-(gx * g.sn) - gy * g.cs) + ghl * (gx * g.cs - gy * g.sn);
      dot = fmaf(wgt, ga, dot);
      if (own) ga_s[slot * PMAX + op] = ga;
    }
    dcx = quad_sum(dcx);
    dcy = quad_sum(dcy);
    dw = quad_sum(dw);
    dh = quad_sum(dh);
    dth = quad_sum(dth);
    dot = quad_sum(dot);
    dcx += __shfl_xor(dcx, 4, 64);
    dcy += __shfl_xor(dcy, 4, 64);
    dw += __shfl_xor(dw, 4, 64);
    dh += __shfl_xor(dh, 4, 64);
    dth += __shfl_xor(dth, 4, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (qok && sub == 0) {
      float* go = grad_off + bqs * dm.off_rs + m * dm.v;
      go[0] = dcx * g.rw / 8.0f;
      go[1] = dcy * g.rh / 8.0f;
      go[2] = g.w_on ? dw * g.rw / 8.0f : 0.0f;
      go[3] = g.h_on ? dh * g.rh / 8.0f : 0.0f;
      if (dm.v == 5) go[4] = dth * (2.0f * 3.14159274f / 16.0f);
    }
    __syncthreads();  // ga_s complete
    if (qok)
      for (int e = sub; e < np; e += 8) grad_logits[bqs * dm.lg_rs + m * np + e] = as[e] * inv * (ga_s[slot * PMAX + e] - dot);
    (void)t;
  }
  if (absmax) {   // the largest |grad_out| this workgroup has seen: one atomic per wave (box_bin_reduce_kernel takes the max of the slots)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) go_max = max(go_max, (unsigned)__shfl_xor((int)go_max, d, 64));
    if (lane == 0 && go_max)
      atomicMax(absmax + 1 + (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z) + wave) % kAbsmaxSlots, go_max);
  }
}

template <int TQY>
__global__ void __launch_bounds__(BT<TQY>::kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
box_bwd_tile_b_kernel(const long long* __restrict__ shapes, const float* __restrict__ ref, const float* __restrict__ off,
                      const float* __restrict__ logits, const float* __restrict__ kidx, const float* __restrict__ grad_out,
                      BoxDims dm, float* __restrict__ grad_value, int* __restrict__ cursor, int2* __restrict__ entries,
                      const int* __restrict__ bin_end, int* __restrict__ overflow, int color) {
  using B = BT<TQY>;
  constexpr int TQX = B::TQX, R = B::R, WINY = B::WINY, WINX = B::WINX, NQ = B::NQ, NC = B::NC, D = B::D, VS = B::VS, WS = B::WS,
                kThreads = B::kThreads, PMAX = bt::PMAX;
  extern __shared__ float lds[];
  float* Wl = lds;                     // W [NC][WS]
  float* GOs = Wl + NC * WS;           // [NQ][VS]
  float* a_s = GOs + NQ * VS;          // [NQ][PMAX]
  float* k_s = a_s + NQ * PMAX;        // [PMAX][2]
  const int Hm = (int)shapes[0], Wm = (int)shapes[1];
  if ((long long)Hm * Wm != dm.s) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 2 * dm.p) k_s[tid] = kidx[tid];
  const int slot = tid >> 3, sub = tid & 7, corner = sub & 3, half = sub >> 2;
  const int m = blockIdx.y, bi = blockIdx.z;
  const int np = dm.p;
  const int tiles_x = (Wm + TQX - 1) / TQX, tiles_y = (Hm + TQY - 1) / TQY;
  constexpr int NCY = (WINY + TQY - 1) / TQY, NCX = (WINX + TQX - 1) / TQX;   // colour classes, as in box_bwd_tile_kernel
  const int cy0 = color >= 0 ? color / NCX : 0, cx0 = color >= 0 ? color % NCX : 0;
  const int sy_ = color >= 0 ? NCY : 1, sx_ = color >= 0 ? NCX : 1;
  const int ctx = (tiles_x - cx0 + sx_ - 1) / sx_, cty = (tiles_y - cy0 + sy_ - 1) / sy_;
  const int ntiles = max(ctx, 0) * max(cty, 0);
  const long long S = dm.s;
  constexpr int EPT = PMAX / 2;
  for (int tj = blockIdx.x; tj < ntiles; tj += gridDim.x) {
    const int tile = (cy0 + sy_ * (tj / ctx)) * tiles_x + cx0 + sx_ * (tj % ctx);
    const int ty0 = (tile / tiles_x) * TQY, tx0 = (tile % tiles_x) * TQX;
    const int wy0 = ty0 - R, wx0 = tx0 - R;
    __syncthreads();  // the previous tile's GEMM is done with GOs / W
    const int qy = ty0 + slot / TQX, qx = tx0 + slot % TQX;
    const bool qok = qy < Hm && qx < Wm;
    const int qyc = min(qy, Hm - 1), qxc = min(qx, Wm - 1);
    const long long bqc = (long long)bi * dm.lq + (long long)qyc * Wm + qxc;
    const long long t = bqc * dm.h + m;   // (only used when qok)
    {
      const float4 go = ld4(grad_out + t * D + sub * 4);
      *reinterpret_cast<float4*>(GOs + slot * VS + sub * 4) = qok ? go : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float lgv[4], rf[5], of[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) lgv[k] = logits[bqc * dm.lg_rs + m * np + min(sub + 8 * k, np - 1)];
    {
      const float* r = ref + bqc * 7;
      rf[0] = r[0], rf[1] = r[1], rf[2] = r[3], rf[3] = r[4], rf[4] = r[6];
#pragma unroll
      for (int k = 0; k < 5; ++k) of[k] = off[bqc * dm.off_rs + m * dm.v + min(k, dm.v - 1)];
    }
    for (int i = tid; i < NC * WS / 4; i += kThreads) reinterpret_cast<float4*>(Wl)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float* as = a_s + slot * PMAX;
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sub + 8 * k < np) mx = fmaxf(mx, lgv[k]);
#pragma unroll
    for (int dlt = 4; dlt > 0; dlt >>= 1) mx = fmaxf(mx, __shfl_xor(mx, dlt, 64));
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sub + 8 * k < np) {
        const float ex = expf(lgv[k] - mx);
        as[sub + 8 * k] = ex;
        den += ex;
      }
#pragma unroll
    for (int dlt = 4; dlt > 0; dlt >>= 1) den += __shfl_xor(den, dlt, 64);
    const float inv = 1.0f / den;
    const BoxGeo g = make_box_v(rf, of, dm.v);
    __syncthreads();

    // ---- the (cell, weight) of this lane's corner of every point, exactly as pass A of the one-kernel form computes them
    int e_cell[EPT];    // window cell (>= 0), -1: nothing to add, -2: outside the window (binned / global path)
    float e_w[EPT];
    const int cyo = corner >> 1, cxo = corner & 1;
    const float sy = cyo ? 1.f : -1.f, ay = cyo ? 0.f : 1.f, sx = cxo ? 1.f : -1.f, ax = cxo ? 0.f : 1.f;
#pragma unroll
    for (int j = 0; j < EPT / 4; ++j) {
      const int op = half + 2 * (4 * j + corner);
      const bool own = qok && op < np;
      const float kxn = own ? k_s[op * 2] : 0.f, kyn = own ? k_s[op * 2 + 1] : 0.f;
      const BoxPx px = box_point(g, kxn, kyn, Hm, Wm);
      const float wgt = own ? as[op] * inv : 0.f;
      const float h_im = px.h_im, w_im = px.w_im;
      const bool inside = own && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hm) && (w_im < (float)Wm);
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const float lh = h_im - (float)h_low, lwf = w_im - (float)w_low;
      const int h_pub = inside ? h_low : -(1 << 20);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 4 * j + c;
        const int hb = quad_bcast_i(h_pub, c), wb = quad_bcast_i(w_low, c);
        const float lhb = quad_bcast_f(lh, c), lwb = quad_bcast_f(lwf, c), wgb = quad_bcast_f(wgt, c);
        const int cy = hb + cyo, cx = wb + cxo;
        const bool ok = (unsigned)cy < (unsigned)Hm && (unsigned)cx < (unsigned)Wm;
        const int ly = cy - wy0, lx = cx - wx0;
        const bool in_win = (unsigned)ly < (unsigned)WINY && (unsigned)lx < (unsigned)WINX;
        e_cell[k] = -1;
        e_w[k] = 0.f;
        if (ok) {
          e_w[k] = wgb * (fmaf(sy, lhb, ay) * fmaf(sx, lwb, ax));
          e_cell[k] = in_win ? ly * WINX + lx : -2;
        }
      }
    }
    // ---- S3 / S4: W[cell][q], the two halves in turn (box_bwd_tile_kernel: why plain read-modify-writes are safe) ------
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        if (half == hsel && e_cell[k] >= 0) Wl[e_cell[k] * WS + slot] += e_w[k];
        asm volatile("" ::: "memory");
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (e_cell[k] == -2) {
        const int pi = half + 2 * k;
        const BoxPx px = box_point(g, k_s[pi * 2], k_s[pi * 2 + 1], Hm, Wm);
        const int cy = (int)floorf(px.h_im) + (corner >> 1), cx = (int)floorf(px.w_im) + (corner & 1);
        const long long bin = ((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m;
        if (cursor) {
          bin_push(cursor, bin_end, overflow, entries, bin, (int)t, e_w[k]);
        } else {
          float* gv = grad_value + bin * D;
#pragma unroll 1
          for (int c = 0; c < D; ++c) unsafeAtomicAdd(gv + c, e_w[k] * GOs[slot * VS + c]);
        }
      }
    }
    __syncthreads();

    // ---- S5: GV = W . GO and the window flush, as in box_bwd_tile_kernel ---------------------------------------------
    {
      const int r16 = lane & 15, kk = lane >> 4;
      constexpr int CBW = B::CBW;
      f32x4 acc[CBW][2];
#pragma unroll
      for (int i = 0; i < CBW; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      float cur[CBW][2][4];
      if (color >= 0) {
#pragma unroll
        for (int i = 0; i < CBW; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cell = 16 * (CBW * wave + i) + 4 * kk + r;
            const int cy = min(max(wy0 + cell / WINX, 0), Hm - 1), cx = min(max(wx0 + cell % WINX, 0), Wm - 1);
            const float* gv = grad_value + (((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m) * D + r16;
            cur[i][0][r] = gv[0];
            cur[i][1][r] = gv[16];
          }
      }
      const float* ap = Wl + (16 * (CBW * wave) + r16) * WS + kk;
      const float* bp = GOs + kk * VS + r16;
#pragma unroll
      for (int ks = 0; ks < NQ / 4; ++ks) {
        const float b0 = bp[4 * ks * VS], b1 = bp[4 * ks * VS + 16];
#pragma unroll
        for (int i = 0; i < CBW; ++i) {
          const float av = ap[i * 16 * WS + 4 * ks];
          acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[i][1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cell = 16 * (CBW * wave + i) + 4 * kk + r;
          const int cy = wy0 + cell / WINX, cx = wx0 + cell % WINX;
          if (cy >= 0 && cy < Hm && cx >= 0 && cx < Wm) {
            float* gv = grad_value + (((long long)bi * S + (long long)cy * Wm + cx) * dm.h + m) * D + r16;
            if (color >= 0) {
              if (acc[i][0][r] != 0.0f) gv[0] = __fadd_rn(cur[i][0][r], acc[i][0][r]);
              if (acc[i][1][r] != 0.0f) gv[16] = __fadd_rn(cur[i][1][r], acc[i][1][r]);
            } else {
              if (acc[i][0][r] != 0.0f) unsafeAtomicAdd(gv, acc[i][0][r]);
              if (acc[i][1][r] != 0.0f) unsafeAtomicAdd(gv + 16, acc[i][1][r]);
            }
          }
        }
    }
  }
}

// ---- binned grad_value (decoder) -------------------------------------------------------------------------
// Decoder queries sample anywhere, so their grad_value contributions cannot be tiled; as float atomics they are
// D atomics per (query, head, point, corner), and on MI355X device-scope atomics are executed past the per-XCD
// L2s (1.0 GB of atomic traffic and 0.75 ms per launch for 2 x 1240 queries).  Instead every corner emits ONE
// 8-byte entry (grad_out row, weight) into the bin of its (scene, cell, head) row:
//   count (1 int atomic per corner) -> exclusive scan -> main kernel writes entries at atomically allocated
//   slots -> one 8-lane group per non-empty bin sums weight * grad_out[row] and owns the output row.
__global__ void __launch_bounds__(256)
box_bin_count_kernel(const long long* __restrict__ shapes, const long long* __restrict__ starts,
                     const float* __restrict__ ref, const float* __restrict__ off, const float* __restrict__ kidx,
                     BoxDims dm, int* __restrict__ counts, int outside_tile_window, int tqy) {
  // one thread per (query, head, level): the box (its trigonometry) once, then the P lattice points -- the same
  // location arithmetic as the backward kernels (make_box / box_point).  outside_tile_window: queries are the cells of
  // the (single) map and only the corners box_bwd_tile_kernel cannot keep in the window of the query's TQY x 8 tile are
  // counted; a box that cannot reach outside its window -- nearly all of them -- costs nothing beyond make_box.
  const long long total = (long long)dm.b * dm.lq * dm.h * dm.l;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int li = (int)(e % dm.l);
  const long long t = e / dm.l;
  const int m = (int)(t % dm.h);
  const long long bq = t / dm.h;
  const int bi = (int)(bq / dm.lq);
  const int H = (int)shapes[li * 2], W = (int)shapes[li * 2 + 1];
  int wy0 = 0, wx0 = 0;
  const int winy = tqy + 2 * bt::R, winx = bt::TQX + 2 * bt::R;
  const BoxGeo g = make_box(ref + bq * 7, off + bq * dm.off_rs + (m * dm.l + li) * dm.v, dm.v);
  if (outside_tile_window) {
    const int q = (int)(bq % dm.lq);
    wy0 = (q / W) / tqy * tqy - bt::R;
    wx0 = (q % W) / bt::TQX * bt::TQX - bt::R;
    // the lattice offsets are below half a box side (|k| < 0.5) in each axis before the rotation, so no sampling point
    // is further than (w + h) / 2 from the box centre in x or in y (w / 2 and h / 2 for an upright box); a box whose
    // whole reach (+ the bilinear neighbour, + 0.01 cell: the per-point positions are rounded 4 more times, ~1e-4 cell
    // at 200 cells) stays inside the window has nothing to count.  Centre and size come from the SAME make_box the
    // backward kernel classifies with.
    const bool upright = g.sn == 0.f;   // the encoder's boxes: no rotation at all, reach = half a side per axis
    const float ex = upright ? 0.5f * g.w : 0.5f * (g.w + g.h), ey = upright ? 0.5f * g.h : 0.5f * (g.w + g.h);
    const float bx = g.cx * (float)W - 0.5f, by = g.cy * (float)H - 0.5f;
    const float rx = ex * (float)W + 0.01f, ry = ey * (float)H + 0.01f;
    if (floorf(bx - rx) >= (float)wx0 && floorf(bx + rx) + 1.f <= (float)(wx0 + winx - 1) &&
        floorf(by - ry) >= (float)wy0 && floorf(by + ry) + 1.f <= (float)(wy0 + winy - 1))
      return;
  }
  int* row = counts + ((long long)bi * dm.s + starts[li]) * dm.h + m;
  for (int pi = 0; pi < dm.p; ++pi) {
    const BoxPx px = box_point(g, kidx[pi * 2], kidx[pi * 2 + 1], H, W);
    const float h_im = px.h_im, w_im = px.w_im;
    if (!((h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W))) continue;
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
#pragma unroll
    for (int cn = 0; cn < 4; ++cn) {
      const int cy = h_low + (cn >> 1), cx = w_low + (cn & 1);
      if (cy >= 0 && cy <= H - 1 && cx >= 0 && cx <= W - 1) {
        if (outside_tile_window && (unsigned)(cy - wy0) < (unsigned)winy && (unsigned)(cx - wx0) < (unsigned)winx) continue;
        atomicAdd(row + ((long long)cy * W + cx) * dm.h, 1);
      }
    }
  }
}

constexpr int kScanTile = 1024;  // elements per workgroup of the 3-kernel exclusive scan

__global__ void __launch_bounds__(256) scan_tiles_kernel(int* __restrict__ data, long long n, int* __restrict__ totals) {
  __shared__ int sm[17];
  const long long base = (long long)blockIdx.x * kScanTile + threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (base + j < n) ? data[base + j] : 0;
  int tot;
  const int pre = block_exclusive_scan(v[0] + v[1] + v[2] + v[3], sm, &tot);
  int run = pre;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) data[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024) scan_totals_kernel(int* __restrict__ totals, int nblk) {
  __shared__ int sm[17];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? totals[i] : 0;
    int tot;
    const int pre = block_exclusive_scan(v, sm, &tot);
    const int carry = carry_s;
    if (i < nblk) totals[i] = carry + pre;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
}

// offsets[i] += totals[tile]; cursor[i] = offsets[i]
__global__ void __launch_bounds__(256) scan_apply_kernel(int* __restrict__ offsets, long long n,
                                                         const int* __restrict__ totals, int* __restrict__ cursor) {
  const long long base = (long long)blockIdx.x * kScanTile + threadIdx.x * 4;
  const int add = totals[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) {
      const int o = offsets[base + j] + add;
      offsets[base + j] = o;
      cursor[base + j] = o;
    }
}

// max |x| of a tensor as float bits (0x7f800000 and above: an Inf / NaN is present) into 63 slots (out[1 + block % 63]: one
// address is one serial atomic unit, ~15 ns per atomic); eight 16-byte loads in flight per thread, one atomic per block.
// (First version: one load per trip and an atomic per wave on ONE word -- 66 us for 72 MB.)
// (kAbsmaxSlots: defined with the tile kernels above)
__global__ void __launch_bounds__(256) absmax_bits_kernel(const float* __restrict__ x, long long n4, unsigned* __restrict__ out) {
  __shared__ unsigned sm[4];
  unsigned m = 0u;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  auto fold = [&](const float4& v) {
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
    m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
  };
  for (; i + 7 * stride < n4; i += 8 * stride) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld4(x + (i + u * stride) * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) fold(v[u]);
  }
  for (; i < n4; i += stride) fold(ld4(x + i * 4));
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, d, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    if (m) atomicMax(out + 1 + blockIdx.x % kAbsmaxSlots, m);
  }
}

// grad_value row `bin` += sum over its entries of weight * grad_out[row]; 8 lanes x float4 per bin.  After the
// main kernel cursor[bin] is the END of the bin, offsets[bin] its start.
// The entries of a bin sit in the order their atomics arrived, which differs run to run; a float accumulation in list
// order would make grad_value (and every gradient upstream of it) differ in the last bits between two identical steps.
// The sum is therefore taken as EXACT 64-bit fixed-point integers at a scale 2^sh at which no product leaves 51 bits and no
// sum of the bin's n products can overflow: every |weight| is at most 1 (a softmax weight times a bilinear weight; 2 is
// the bound used) and every |grad_out| at most the largest of the call (the maximum over `gmax_bits[1..63]`,
// absmax_bits_kernel), so the bound needs no pass over the bin (round 4: a pass over the bin's 8-byte entries for its largest
// |weight|; before that a gather pass for its exact largest product -- the gathers are the cost of this kernel, and it
// doubles as the boxes grow and most corners of the encoder leave their query tile's window).  Each product is exact in double
// (24 x 24 bits), scaling by 2^sh is exact, and ONE fused multiply-add with 1.5 * 2^52 rounds it to an integer that sits in
// the low mantissa bits (|x| < 2^51): two fp64 instructions per product instead of ldexp + a software double -> int64
// conversion.  Integer addition is associative, and the row receives the total rounded once: exact to 2^-(50 - log2 n) of
// the bound, i.e. to ~1e-12 of the call's largest gradient element.
__global__ void __launch_bounds__(256)
box_bin_reduce_kernel(const int* __restrict__ offsets, const int* __restrict__ cursor, const int2* __restrict__ entries,
                      const float* __restrict__ grad_out, long long nbins, const unsigned* __restrict__ gmax_bits,
                      float* __restrict__ grad_value) {
  const int c4 = (threadIdx.x & 7) * 4;
  unsigned gbits = 0u;   // (the 63 slots of absmax_bits_kernel, after the overflow word of the flag block)
  for (int i = 1; i <= kAbsmaxSlots; ++i) gbits = max(gbits, gmax_bits[i]);
  const bool finite = gbits < 0x7f800000u;
  int ex = 0;
  frexpf(fminf(__uint_as_float(gbits) * 2.0000002f, 3.0e38f), &ex);   // every |product| < 2^ex
  constexpr double kMagic = 6755399441055744.0;   // 1.5 * 2^52: x + kMagic holds round(x) in its low mantissa bits for |x| < 2^51
  const long long magic_bits = __double_as_longlong(kMagic);
  for (long long bin = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); bin < nbins; bin += (long long)gridDim.x * 32) {
    const int s = offsets[bin], e = min(cursor[bin], offsets[bin + 1]);  // offsets has nbins + 1 entries
    if (s == e) continue;
    float4 acc = ld4(grad_value + bin * 32 + c4);
    if (finite) {
      int ln = 0;
      while ((1 << ln) < e - s + 1) ++ln;    // n + 1 <= 2^ln
      const int sh = min(50, 62 - ln) - ex;  // |product * 2^sh| < 2^50 (one rounding bit to spare), |sum| < 2^62
      const double scale = ldexp(1.0, sh);
      long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (int i0 = s; i0 < e; i0 += 4) {   // four entries (and their grad_out rows) in flight
        int2 en[4];
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = entries[min(i0 + u, e - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = ld4(grad_out + (long long)en[u].x * 32 + c4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double w = i0 + u < e ? (double)__int_as_float(en[u].y) : 0.0;   // (past the end: + 0)
          a0 += __double_as_longlong(fma(w * (double)g[u].x, scale, kMagic)) - magic_bits;
          a1 += __double_as_longlong(fma(w * (double)g[u].y, scale, kMagic)) - magic_bits;
          a2 += __double_as_longlong(fma(w * (double)g[u].z, scale, kMagic)) - magic_bits;
          a3 += __double_as_longlong(fma(w * (double)g[u].w, scale, kMagic)) - magic_bits;
        }
      }
      acc.x = __fadd_rn(acc.x, (float)ldexp((double)a0, -sh));
      acc.y = __fadd_rn(acc.y, (float)ldexp((double)a1, -sh));
      acc.z = __fadd_rn(acc.z, (float)ldexp((double)a2, -sh));
      acc.w = __fadd_rn(acc.w, (float)ldexp((double)a3, -sh));
    } else {   // Inf / NaN among the gradients: so are the sums they reach, in any order
      for (int i = s; i < e; ++i) {
        const int2 en = entries[i];
        const float4 g = ld4(grad_out + (long long)en.x * 32 + c4);
        const float w = __int_as_float(en.y);
        acc.x = fmaf(w, g.x, acc.x);
        acc.y = fmaf(w, g.y, acc.y);
        acc.z = fmaf(w, g.z, acc.z);
        acc.w = fmaf(w, g.w, acc.w);
      }
    }
    *reinterpret_cast<float4*>(grad_value + bin * 32 + c4) = acc;
  }
}

struct BinPlan {
  long long nbins, nentries;
  int ntiles;
  size_t off_flag, off_offsets, off_cursor, off_totals, off_entries, bytes;
};

// workspace: [overflow counter (int, first word)] [offsets: nbins + 1] [cursor: nbins + 1] [scan totals] [entries]
BinPlan bin_plan(int b, int s, int h, int l, int lq, int p) {
  BinPlan pl;
  pl.nbins = (long long)b * s * h;
  pl.nentries = (long long)b * lq * h * l * p * 4;
  pl.ntiles = (int)ceil_div(pl.nbins + 1, kScanTile);
  size_t o = 0;
  pl.off_flag = o;
  o += 256;
  pl.off_offsets = o;
  o += align_up(sizeof(int) * (size_t)(pl.nbins + 1), 256);
  pl.off_cursor = o;
  o += align_up(sizeof(int) * (size_t)(pl.nbins + 1), 256);
  pl.off_totals = o;
  o += align_up(sizeof(int) * (size_t)pl.ntiles, 256);
  pl.off_entries = o;
  o += align_up(sizeof(int2) * (size_t)pl.nentries, 256);
  pl.bytes = o;
  return pl;
}

// count -> exclusive scan over nbins + 1 counters (the last one stays 0, so offsets[nbins] = number of entries and
// offsets[bin + 1] ends every bin); cursor = copy of offsets
int bin_scan(const BinPlan& pl, void* ws, hipStream_t st) {
  char* base = static_cast<char*>(ws);
  int* offs = reinterpret_cast<int*>(base + pl.off_offsets);
  int* cursor = reinterpret_cast<int*>(base + pl.off_cursor);
  int* totals = reinterpret_cast<int*>(base + pl.off_totals);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(pl.ntiles), dim3(256), 0, st, offs, pl.nbins + 1, totals);
  hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, st, totals, pl.ntiles);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(pl.ntiles), dim3(256), 0, st, offs, pl.nbins + 1, totals, cursor);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

// count_here = false: only the pointers and the memset -- the caller's pass A counts the corners and takes the largest
// |grad_out| along (box_bwd_tile_a_kernel), then calls bin_scan
int bin_prepare(const BinPlan& pl, void* ws, const long long* shapes, const long long* starts, const float* ref,
                const float* off, const float* kidx, const BoxDims& dm, int outside_tile_window, int tqy, hipStream_t st,
                int** offs, int** cursor, int2** entries, int** overflow, const float* grad_out, long long grad_out_floats,
                bool count_here = true) {
  char* base = static_cast<char*>(ws);
  *overflow = reinterpret_cast<int*>(base + pl.off_flag);
  *offs = reinterpret_cast<int*>(base + pl.off_offsets);
  *cursor = reinterpret_cast<int*>(base + pl.off_cursor);
  *entries = reinterpret_cast<int2*>(base + pl.off_entries);
  // flag + offsets are adjacent: one memset
  EFG_HIP_TRY(hipMemsetAsync(base + pl.off_flag, 0, 256 + sizeof(int) * (size_t)(pl.nbins + 1), st));
  if (!count_here) return EFG_OK;
  // words 1..63 of the (zeroed) flag block: largest |grad_out| of the call as float bits (the scale of box_bin_reduce_kernel)
  hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(ceil_div(grad_out_floats / 4, 2048), 1), 1024)),
                     dim3(256), 0, st, grad_out, grad_out_floats / 4, reinterpret_cast<unsigned*>(*overflow));
  const long long nboxes = (long long)dm.b * dm.lq * dm.h * dm.l;
  hipLaunchKernelGGL(box_bin_count_kernel, dim3((unsigned)ceil_div(nboxes, 256)), dim3(256), 0, st, shapes, starts, ref,
                     off, kidx, dm, *offs, outside_tile_window, tqy);
  EFG_LAUNCH_CHECK();
  return bin_scan(pl, ws, st);
}

constexpr long long kBinMinEntries = 200000;  // below this the extra launches cost more than the atomics

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
  *dm = BoxDims{b, s, h, d, l, lq, p, v, lp, sh, h * l * v, h * l * p};
  return EFG_OK;
}

}  // namespace
}  // namespace efg

using namespace efg;

namespace {
// row strides of the offsets / logits matrices (0: each is a dense matrix of its own)
int set_strides(BoxDims* dm, int off_rs, int lg_rs) {
  if (off_rs) {
    EFG_CHECK_ARG(off_rs >= dm->h * dm->l * dm->v, "box_attn_fused: offsets row stride %d below the row width %d", off_rs, dm->h * dm->l * dm->v);
    dm->off_rs = off_rs;
  }
  if (lg_rs) {
    EFG_CHECK_ARG(lg_rs >= dm->h * dm->l * dm->p, "box_attn_fused: logits row stride %d below the row width %d", lg_rs, dm->h * dm->l * dm->p);
    dm->lg_rs = lg_rs;
  }
  return EFG_OK;
}
}  // namespace

extern "C" int efg_box_attn_fused_forward_strided_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                                      const float* ref_windows, const float* offsets, int off_row_stride,
                                                      const float* logits, int logit_row_stride, const float* kernel_indices,
                                                      int b, int s, int h, int d, int l, int lq, int p, int v, float* out,
                                                      void* stream) {
  BoxDims dm;
  if (int rc = check(b, s, h, d, l, lq, p, v, &dm)) return rc;
  if (int rc = set_strides(&dm, off_row_stride, logit_row_stride)) return rc;
  const long long total = (long long)b * lq * h;
  if (total == 0) return EFG_OK;
  const int pairs_per_block = 4 * (64 / dm.lp);
  hipLaunchKernelGGL(box_fwd_kernel, dim3((unsigned)ceil_div(total, pairs_per_block)), dim3(256), 0,
                     (hipStream_t)stream, value, (const long long*)shapes, (const long long*)level_start, ref_windows,
                     offsets, logits, kernel_indices, dm, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_box_attn_fused_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                              const float* ref_windows, const float* offsets, const float* logits,
                                              const float* kernel_indices, int b, int s, int h, int d, int l, int lq,
                                              int p, int v, float* out, void* stream) {
  return efg_box_attn_fused_forward_strided_f32(value, shapes, level_start, ref_windows, offsets, 0, logits, 0, kernel_indices, b, s, h,
                                                d, l, lq, p, v, out, stream);
}

extern "C" size_t efg_box_attn_fused_backward_workspace_bytes(int b, int s, int h, int l, int lq, int p) {
  if (b < 0 || s < 0 || h < 1 || l < 1 || lq < 0 || p < 1) return 0;
  return bin_plan(b, s, h, l, lq, p).bytes;
}

extern "C" int efg_box_attn_fused_backward_strided_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                                       const float* ref_windows, const float* offsets, int off_row_stride,
                                                       const float* logits, int logit_row_stride, const float* kernel_indices,
                                                       const float* grad_out, int b, int s, int h, int d, int l, int lq, int p,
                                                       int v, float* grad_value, float* grad_offsets, float* grad_logits,
                                                       void* ws, size_t ws_bytes, void* stream) {
  BoxDims dm;
  if (int rc = check(b, s, h, d, l, lq, p, v, &dm)) return rc;
  if (int rc = set_strides(&dm, off_row_stride, logit_row_stride)) return rc;   // (the gradients have the strides of their matrices)
  EFG_CHECK_ARG(d == 32, "box_attn_fused backward: head dim 32 only (got %d); use the unfused op", d);
  const long long total = (long long)b * lq * h;
  if (total == 0) return EFG_OK;
  if (l == 1 && s == lq && s >= 1024 && h <= 65535 && b <= 65535) {
    // encoder self-attention (queries on the value map): fp64 LDS window per 8x8 query tile.  H, W are
    // device-side, so the launch is sized for a square map and the kernel strides over the tiles.
    const int side = (int)std::ceil(std::sqrt((double)s));
    // tile shape: 4 x 8 queries (the 8 x 8 tile of round 2 -- BT<8>, one workgroup per CU -- was retired in round 6)
    constexpr int tqy = 4;
    const unsigned tiles_sq = (unsigned)(((side + tqy - 1) / tqy) * ((side + 7) / 8));
    // workgroups along x of the tile kernel: it strides over the tiles and fetches one tile ahead (64 x h x b workgroups,
    // two rounds of the 512 resident ones: 507 -> 490 us against one workgroup per tile)
    constexpr int gx_env = 0;
    const unsigned tile_gx = std::min<unsigned>((unsigned)(gx_env > 0 ? gx_env : 64), tiles_sq);
    if (l * p <= bt::PMAX) {
      // corners that leave the tile's window are binned per (cell, head) row when a workspace is given (see the kernel)
      const BinPlan pl = bin_plan(b, s, h, l, lq, p);
      hipStream_t st = (hipStream_t)stream;
      const bool binned = ws != nullptr && pl.nentries < (1ll << 31) && pl.nbins + 1 < (1ll << 31);
      int *cursor = nullptr, *offs = nullptr, *overflow = nullptr;
      int2* entries = nullptr;
      // EFG_BOX_SPLIT (default 1): pass A and pass B as two kernels at four waves per SIMD (see box_bwd_tile_a_kernel);
      // 0 = the one-kernel form.
      const int split_env = getenv("EFG_BOX_SPLIT") ? atoi(getenv("EFG_BOX_SPLIT")) : 1;   // (read per call: tests flip it in-process)
      const bool split = split_env != 0;
      if (binned) {
        EFG_CHECK_ARG(ws_bytes >= pl.bytes, "box_attn_fused backward: workspace too small (%zu < %zu)", ws_bytes, pl.bytes);
        // split: pass A counts the out-of-window corners and takes the largest |grad_out| along (two launches less per call)
        if (int rc = bin_prepare(pl, ws, (const long long*)shapes, (const long long*)level_start, ref_windows, offsets,
                                 kernel_indices, dm, 1, tqy, st, &offs, &cursor, &entries, &overflow, grad_out,
                                 (long long)dm.b * dm.lq * dm.h * dm.d, /*count_here=*/!split))
          return rc;
      }
      // EFG_BOX_DETERMINISTIC (default 1): the tiles in NCY x NCX colour classes whose windows never overlap, one launch
      // per class with a plain read-modify-write flush -- grad_value is the same bits run to run (with the binned
      // out-of-window corners, whose reduction is order-independent).  0: one launch, float atomics.
      static const int det_env = getenv("EFG_BOX_DETERMINISTIC") ? atoi(getenv("EFG_BOX_DETERMINISTIC")) : 1;
      const bool colored = det_env != 0 && binned;
      const int ncolors = colored ? ((BT<4>::WINY + 3) / 4) * ((BT<4>::WINX + BT<4>::TQX - 1) / BT<4>::TQX) : 1;
      // (workgroups along x of a colour launch; measured 16: +1.1 ms, 32: +0.35 ms, 64: +0.1 ms per step against the atomic launch)
      constexpr int cgx_env = 64;
      const unsigned gx_launch = colored ? std::min<unsigned>(tile_gx, (unsigned)std::max(cgx_env, 1)) : tile_gx;
      if (split) {
        EFG_ALLOW_DYNAMIC_LDS(box_bwd_tile_a_kernel<4>, bt::lds_bytes_a<4>());
        hipLaunchKernelGGL(box_bwd_tile_a_kernel<4>, dim3(tile_gx, h, b), dim3(BT<4>::kThreads), bt::lds_bytes_a<4>(), st, value,
                           (const long long*)shapes, ref_windows, offsets, logits, kernel_indices, grad_out, dm, grad_offsets,
                           grad_logits, binned ? offs : nullptr, binned ? reinterpret_cast<unsigned*>(overflow) : nullptr);
        EFG_LAUNCH_CHECK();
        if (binned)
          if (int rc = bin_scan(pl, ws, st)) return rc;
        EFG_ALLOW_DYNAMIC_LDS(box_bwd_tile_b_kernel<4>, bt::lds_bytes_b<4>());
      }
      for (int col = 0; col < ncolors; ++col) {
        const int color = colored ? col : -1;
        if (split) {
          hipLaunchKernelGGL(box_bwd_tile_b_kernel<4>, dim3(gx_launch, h, b), dim3(BT<4>::kThreads), bt::lds_bytes_b<4>(), st,
                             (const long long*)shapes, ref_windows, offsets, logits, kernel_indices, grad_out, dm, grad_value,
                             cursor, entries, offs ? offs + 1 : nullptr, overflow, color);
        } else {
          EFG_ALLOW_DYNAMIC_LDS(box_bwd_tile_kernel<4>, bt::lds_bytes<4>());
          hipLaunchKernelGGL(box_bwd_tile_kernel<4>, dim3(gx_launch, h, b), dim3(BT<4>::kThreads), bt::lds_bytes<4>(), st, value,
                             (const long long*)shapes, ref_windows, offsets, logits, kernel_indices, grad_out, dm, grad_value,
                             grad_offsets, grad_logits, cursor, entries, offs ? offs + 1 : nullptr, overflow, color);
        }
      }
      if (binned) {
        EFG_LAUNCH_CHECK();
        const unsigned blocks = (unsigned)std::min<long long>(ceil_div(pl.nbins, 32), 16384);
        hipLaunchKernelGGL(box_bin_reduce_kernel, dim3(blocks), dim3(256), 0, st, offs, cursor, entries, grad_out, pl.nbins,
                           reinterpret_cast<const unsigned*>(overflow), grad_value);
      }
    }
    else
      hipLaunchKernelGGL((box_bwd_kernel<32, true>), dim3(tiles_sq, h, b), dim3(256), 0, (hipStream_t)stream,
                         value, (const long long*)shapes, (const long long*)level_start, ref_windows, offsets, logits,
                         kernel_indices, grad_out, dm, grad_value, grad_offsets, grad_logits, nullptr, nullptr, nullptr, nullptr);
  } else {
    const BinPlan pl = bin_plan(b, s, h, l, lq, p);
    hipStream_t st = (hipStream_t)stream;
    // small problems (below kBinMinEntries corners the extra launches cost more than the float atomics) are binned too
    // unless EFG_BOX_DETERMINISTIC=0: the atomics' arrival order would make grad_value differ run to run
    static const int det_dec = getenv("EFG_BOX_DETERMINISTIC") ? atoi(getenv("EFG_BOX_DETERMINISTIC")) : 1;
    const bool binned = ws != nullptr && (det_dec != 0 || pl.nentries >= kBinMinEntries) && pl.nentries < (1ll << 31) &&
                        pl.nbins + 1 < (1ll << 31);
    int *cursor = nullptr, *offs = nullptr, *overflow = nullptr;
    int2* entries = nullptr;
    if (binned) {
      EFG_CHECK_ARG(ws_bytes >= pl.bytes, "box_attn_fused backward: workspace too small (%zu < %zu)", ws_bytes, pl.bytes);
      if (int rc = bin_prepare(pl, ws, (const long long*)shapes, (const long long*)level_start, ref_windows, offsets,
                               kernel_indices, dm, 0, 8, st, &offs, &cursor, &entries, &overflow, grad_out,
                               (long long)dm.b * dm.lq * dm.h * dm.d))
        return rc;
    } else if (ws != nullptr && ws_bytes >= sizeof(int)) {
      EFG_HIP_TRY(hipMemsetAsync(ws, 0, sizeof(int), st));  // the overflow word is defined whenever a workspace is given
    }
    hipLaunchKernelGGL((box_bwd_kernel<32, false>), dim3((unsigned)ceil_div(total, 32)), dim3(256), 0, st, value,
                       (const long long*)shapes, (const long long*)level_start, ref_windows, offsets, logits,
                       kernel_indices, grad_out, dm, grad_value, grad_offsets, grad_logits, cursor, entries, offs ? offs + 1 : nullptr,
                       overflow);
    if (binned) {
      EFG_LAUNCH_CHECK();
      const unsigned blocks = (unsigned)std::min<long long>(ceil_div(pl.nbins, 32), 16384);
      hipLaunchKernelGGL(box_bin_reduce_kernel, dim3(blocks), dim3(256), 0, st, offs, cursor, entries, grad_out,
                         pl.nbins, reinterpret_cast<const unsigned*>(overflow), grad_value);
    }
  }
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_box_attn_fused_backward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                               const float* ref_windows, const float* offsets, const float* logits,
                                               const float* kernel_indices, const float* grad_out, int b, int s, int h,
                                               int d, int l, int lq, int p, int v, float* grad_value,
                                               float* grad_offsets, float* grad_logits, void* ws, size_t ws_bytes,
                                               void* stream) {
  return efg_box_attn_fused_backward_strided_f32(value, shapes, level_start, ref_windows, offsets, 0, logits, 0, kernel_indices,
                                                 grad_out, b, s, h, d, l, lq, p, v, grad_value, grad_offsets, grad_logits, ws,
                                                 ws_bytes, stream);
}
