// Detection losses of Voxel-DETR / ConQueR as fused kernels ($CQ/modules/matcher.py:40-80, $CQ/losses.py:26-108).
//
// The reference (and the batched PyTorch form in detection3d/losses.py) evaluates the matching cost and the
// set-prediction losses with ~40 tiny elementwise kernels per call and as many again backward -- several hundred
// launches of a few microseconds each per step, i.e. pure launch latency on both the host and the GPU.  Each
// family is ONE kernel here (plus one for its gradient):
//   match_cost_kernel      cost[p, q, g] = w_b*L1(xyz,lwh) + w_c*(focal class cost) + w_g*(-GIoU3D) + w_r*|d rad|
//   focal_sum_kernel       per-layer sum of the sigmoid focal loss against a class-index target (-1 = background)
//   focal_grad_kernel      its gradient w.r.t. the logits
//   box_loss_kernel        per-layer sums of L1(xyz,lwh), 1 - GIoU3D, |d rad| over matched (prediction, target) pairs
//   box_loss_grad_kernel   gradient w.r.t. the predicted boxes (a query is matched at most once: plain stores)
// Reductions are one workgroup per layer with a fixed tree: deterministic.
#include "common.h"

#include <algorithm>

namespace efg {
namespace {

__device__ __forceinline__ float nan_to_num(float v) {  // torch.nan_to_num defaults
  if (v != v) return 0.0f;
  if (v == INFINITY) return 3.4028234663852886e38f;
  if (v == -INFINITY) return -3.4028234663852886e38f;
  return v;
}

// axis-aligned 3-D GIoU of (centre, size) boxes, as utils.paired_box3d_giou on box_cxcyczlwh_to_xyxyxy
__device__ __forceinline__ float giou3d(const float* s, const float* t) {
  float inter = 1.f, vol = 1.f, v1 = 1.f, v2 = 1.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float slo = nan_to_num(s[k] - 0.5f * s[3 + k]), shi = nan_to_num(s[k] + 0.5f * s[3 + k]);
    const float tlo = nan_to_num(t[k] - 0.5f * t[3 + k]), thi = nan_to_num(t[k] + 0.5f * t[3 + k]);
    v1 *= shi - slo;
    v2 *= thi - tlo;
    inter *= fmaxf(fminf(shi, thi) - fmaxf(slo, tlo), 0.f);
    vol *= fmaxf(fmaxf(shi, thi) - fminf(slo, tlo), 0.f);
  }
  const float uni = v1 + v2 - inter;
  return inter / uni - (vol - uni) / vol;
}

struct CostW {
  float w_class, w_bbox, w_giou, w_rad, alpha, gamma;
};

__global__ void __launch_bounds__(256)
match_cost_kernel(const float* __restrict__ logits, const float* __restrict__ boxes, const long long* __restrict__ tgt_labels,
                  const float* __restrict__ tgt_boxes, int P, int B, int Q, int C, int G, CostW w, float* __restrict__ cost) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)P * Q * G) return;
  const int g = (int)(e % G);
  const long long pq = e / G;
  const int q = (int)(pq % Q), p = (int)(pq / Q), b = p % B;
  const float* bx = boxes + ((long long)p * Q + q) * 7;
  const float* tb = tgt_boxes + ((long long)b * G + g) * 7;
  const int lab = (int)tgt_labels[(long long)b * G + g];
  const float x = logits[((long long)p * Q + q) * C + lab];
  const float pr = 1.0f / (1.0f + expf(-x));
  const float neg = (1.f - w.alpha) * powf(pr, w.gamma) * (-logf(1.f - pr + 1e-8f));
  const float pos = w.alpha * powf(1.f - pr, w.gamma) * (-logf(pr + 1e-8f));
  float l1 = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) l1 += fabsf(bx[k] - tb[k]);
  const float rad = fabsf(bx[6] - tb[6]);
  cost[e] = w.w_bbox * l1 + w.w_class * (pos - neg) + w.w_giou * (-giou3d(bx, tb)) + w.w_rad * rad;
}

// ---- focal loss ------------------------------------------------------------------------------------
// efg/modeling/losses/focal_loss.py:5-45 with targets = one-hot(tcls) (tcls = -1: all zero)
__device__ __forceinline__ float focal_elem(float x, bool t, float alpha, float gamma) {
  const float p = 1.0f / (1.0f + expf(-x));
  const float ce = fmaxf(x, 0.f) - (t ? x : 0.f) + log1pf(expf(-fabsf(x)));
  const float pt = t ? p : 1.f - p;
  // (gamma == 2, the reference's value: ATen evaluates pow(x, 2.0) as x * x too; the general powf was half of this kernel)
  const float one_m = 1.f - pt;
  float loss = ce * (gamma == 2.0f ? one_m * one_m : powf(one_m, gamma));
  if (alpha >= 0.f) loss *= t ? alpha : 1.f - alpha;
  return loss;
}

__device__ __forceinline__ float focal_elem_grad(float x, bool t, float alpha, float gamma) {
  // d/dx [ a_t * ce * (1 - p_t)^gamma ],  dp/dx = p (1 - p)
  const float p = 1.0f / (1.0f + expf(-x));
  const float ce = fmaxf(x, 0.f) - (t ? x : 0.f) + log1pf(expf(-fabsf(x)));
  const float pt = t ? p : 1.f - p;
  const float one_m = 1.f - pt;
  const float dce = p - (t ? 1.f : 0.f);                 // d ce / dx
  const float dpt = (t ? 1.f : -1.f) * p * (1.f - p);    // d p_t / dx
  float g = gamma == 2.0f ? dce * (one_m * one_m) - ce * 2.0f * one_m * dpt
                          : dce * powf(one_m, gamma) - ce * gamma * powf(one_m, gamma - 1.f) * dpt;
  if (alpha >= 0.f) g *= t ? alpha : 1.f - alpha;
  return g;
}

__device__ __forceinline__ float block_sum_1024(float v, float* sm) {  // blockDim.x == 1024
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 16) r = sm[threadIdx.x];
  if (threadIdx.x < 64) {
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) r += __shfl_xor(r, d, 64);
  }
  __syncthreads();
  return r;  // valid in thread 0
}

// logits [L][N][C], tcls int32 [L][N]; out[l] = sum / denom
__global__ void __launch_bounds__(1024)
focal_sum_kernel(const float* __restrict__ logits, const int* __restrict__ tcls, long long N, int C, float alpha,
                 float gamma, const float* __restrict__ denom, float* __restrict__ out) {
  __shared__ float sm[16];
  const int l = blockIdx.x;
  const float* lg = logits + (long long)l * N * C;
  const int* tc = tcls + (long long)l * N;
  float acc = 0.f;
  const long long total = N * C;
  // (sixteen elements in flight, added in the same order: the encoder's 70 688 tokens are ONE workgroup's 69 elements per
  // thread, i.e. round trips -- four in flight made this kernel 61 us)
  constexpr int kFly = 16;
  for (long long e0 = threadIdx.x; e0 < total; e0 += kFly * 1024) {
    float x[kFly];
    bool hit[kFly], on[kFly];
#pragma unroll
    for (int u = 0; u < kFly; ++u) {
      const long long e = e0 + 1024ll * u;
      on[u] = e < total;
      const long long ec = on[u] ? e : 0;
      x[u] = lg[ec];
      hit[u] = tc[ec / C] == (int)(ec % C);
    }
#pragma unroll
    for (int u = 0; u < kFly; ++u)
      if (on[u]) acc += focal_elem(x[u], hit[u], alpha, gamma);
  }
  const float s = block_sum_1024(acc, sm);
  if (threadIdx.x == 0) out[l] = s / denom[0];
}

// The same sum over SPLIT workgroups per layer (the encoder's 70 688 tokens on one workgroup were 88 us of one CU's
// transcendental throughput): workgroup (l, s) sums the elements  s * 1024 + t + k * (1024 * split)  of layer l -- a thread's
// elements in ascending order, as above --, publishes its partial and takes a ticket; the workgroup that draws the layer's last
// ticket adds the partials in s order and stores 0 back into the slots (efg::ticket_slots: zero at rest).  Deterministic: the
// ticket decides WHO adds, not the order.   slots: [layers] counters | [layers * split] partial sums (float bits)
__global__ void __launch_bounds__(1024)
focal_sum_split_kernel(const float* __restrict__ logits, const int* __restrict__ tcls, long long N, int C, float alpha,
                       float gamma, const float* __restrict__ denom, float* __restrict__ out, unsigned* __restrict__ slots) {
  __shared__ float sm[16];
  __shared__ int s_last;
  const int l = blockIdx.x, split = gridDim.y, sidx = blockIdx.y;
  const float* lg = logits + (long long)l * N * C;
  const int* tc = tcls + (long long)l * N;
  unsigned* counter = slots + l;
  float* partial = reinterpret_cast<float*>(slots + gridDim.x) + (long long)l * split;
  const long long total = N * C, stride = 1024ll * split;
  float acc = 0.f;
  constexpr int kFly = 8;
  for (long long e0 = (long long)sidx * 1024 + threadIdx.x; e0 < total; e0 += kFly * stride) {
    float x[kFly];
    bool hit[kFly], on[kFly];
#pragma unroll
    for (int u = 0; u < kFly; ++u) {
      const long long e = e0 + stride * u;
      on[u] = e < total;
      const long long ec = on[u] ? e : 0;
      x[u] = lg[ec];
      hit[u] = tc[ec / C] == (int)(ec % C);
    }
#pragma unroll
    for (int u = 0; u < kFly; ++u)
      if (on[u]) acc += focal_elem(x[u], hit[u], alpha, gamma);
  }
  const float s = block_sum_1024(acc, sm);
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + sidx, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (unsigned)split - 1;
  }
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  float t = 0.f;
  for (int i = 0; i < split; ++i) {
    t += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(partial + i, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  out[l] = t / denom[0];
}

__global__ void __launch_bounds__(256)
focal_grad_kernel(const float* __restrict__ logits, const int* __restrict__ tcls, long long N, int C, int L, float alpha,
                  float gamma, const float* __restrict__ denom, const float* __restrict__ gout, float* __restrict__ glogits) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)L * N * C) return;
  const int c = (int)(e % C);
  const long long li = e / C;
  const int l = (int)(li / N);
  glogits[e] = gout[l] / denom[0] * focal_elem_grad(logits[e], tcls[li] == c, alpha, gamma);
}

// ---- box losses over matched pairs ---------------------------------------------------------------------
// boxes [L][B][Q][7]; pair i: prediction (l_i, b_i, q_i), target tgt[b_i][g_i]; out[l][0..2] = (sum L1 of the 6
// box numbers, sum (1 - GIoU), sum |d rad|) / denom
__global__ void __launch_bounds__(1024)
box_loss_kernel(const float* __restrict__ boxes, const float* __restrict__ tgt, const long long* __restrict__ li,
                const long long* __restrict__ bi, const long long* __restrict__ qi, const long long* __restrict__ gi,
                long long n, int B, int Q, int G, const float* __restrict__ denom, float* __restrict__ out) {
  __shared__ float sm[16];
  const int l = blockIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) {
    if (li[i] != l || qi[i] < 0 || qi[i] >= Q) continue;  // unmatched column (infeasible assignment): no pair
    const float* s = boxes + ((li[i] * B + bi[i]) * Q + qi[i]) * 7;
    const float* t = tgt + (bi[i] * G + gi[i]) * 7;
    float l1 = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) l1 += fabsf(s[k] - t[k]);
    a0 += l1;
    a1 += 1.f - giou3d(s, t);
    a2 += fabsf(s[6] - t[6]);
  }
  const float s0 = block_sum_1024(a0, sm), s1 = block_sum_1024(a1, sm), s2 = block_sum_1024(a2, sm);
  if (threadIdx.x == 0) {
    out[l * 3 + 0] = s0 / denom[0];
    out[l * 3 + 1] = s1 / denom[0];
    out[l * 3 + 2] = s2 / denom[0];
  }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// gout [L][3]; gboxes [L][B][Q][7] must be zero-filled (only matched rows are written; a row is matched once)
__global__ void __launch_bounds__(256)
box_loss_grad_kernel(const float* __restrict__ boxes, const float* __restrict__ tgt, const long long* __restrict__ li,
                     const long long* __restrict__ bi, const long long* __restrict__ qi, const long long* __restrict__ gi,
                     long long n, int B, int Q, int G, const float* __restrict__ denom, const float* __restrict__ gout,
                     float* __restrict__ gboxes) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || qi[i] < 0 || qi[i] >= Q) return;  // unmatched column: nothing to write (never index out of range)
  const long long row = ((li[i] * B + bi[i]) * Q + qi[i]) * 7;
  const float* s = boxes + row;
  const float* t = tgt + (bi[i] * G + gi[i]) * 7;
  const float inv = 1.0f / denom[0];
  const float g_l1 = gout[li[i] * 3 + 0] * inv, g_gi = gout[li[i] * 3 + 1] * inv, g_rd = gout[li[i] * 3 + 2] * inv;
  float g[7];
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = g_l1 * sgn(s[k] - t[k]);
  g[6] = g_rd * sgn(s[6] - t[6]);
  // loss_giou = 1 - giou = 2 - inter/union - union/vol
  float slo[3], shi[3], tlo[3], thi[3], ik[3], ek[3];
  float inter = 1.f, vol = 1.f, v1 = 1.f, v2 = 1.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    slo[k] = s[k] - 0.5f * s[3 + k];
    shi[k] = s[k] + 0.5f * s[3 + k];
    tlo[k] = t[k] - 0.5f * t[3 + k];
    thi[k] = t[k] + 0.5f * t[3 + k];
    v1 *= shi[k] - slo[k];
    v2 *= thi[k] - tlo[k];
    ik[k] = fmaxf(fminf(shi[k], thi[k]) - fmaxf(slo[k], tlo[k]), 0.f);
    ek[k] = fmaxf(fmaxf(shi[k], thi[k]) - fminf(slo[k], tlo[k]), 0.f);
    inter *= ik[k];
    vol *= ek[k];
  }
  const float uni = v1 + v2 - inter;
  // d loss / d inter, d union, d vol
  const float d_inter_direct = -1.f / uni, d_uni = inter / (uni * uni) - 1.f / vol, d_vol = uni / (vol * vol);
  const float d_inter = d_inter_direct - d_uni;  // union = v1 + v2 - inter
  const float d_v1 = d_uni;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    float g_hi = 0.f, g_lo = 0.f;
    // vol1 = prod (shi - slo)
    const float dv1 = d_v1 * (shi[k1] - slo[k1]) * (shi[k2] - slo[k2]);
    g_hi += dv1;
    g_lo -= dv1;
    // inter_k = clamp(min(shi, thi) - max(slo, tlo), 0)
    if (ik[k] > 0.f) {
      const float di = d_inter * ik[k1] * ik[k2];
      if (shi[k] < thi[k]) g_hi += di; else if (shi[k] == thi[k]) g_hi += 0.5f * di;
      if (slo[k] > tlo[k]) g_lo -= di; else if (slo[k] == tlo[k]) g_lo -= 0.5f * di;
    }
    // enc_k = clamp(max(shi, thi) - min(slo, tlo), 0)
    if (ek[k] > 0.f) {
      const float de = d_vol * ek[k1] * ek[k2];
      if (shi[k] > thi[k]) g_hi += de; else if (shi[k] == thi[k]) g_hi += 0.5f * de;
      if (slo[k] < tlo[k]) g_lo -= de; else if (slo[k] == tlo[k]) g_lo -= 0.5f * de;
    }
    g[k] += g_gi * (g_hi + g_lo);
    g[3 + k] += g_gi * 0.5f * (g_hi - g_lo);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) gboxes[row + k] = g[k];
}


// ---- iterative box refinement of the detection heads ($CQ/heads.py:76-79, $CQ/transformer.py:60-81) -----------------
//   out = sigmoid(delta + inverse_sigmoid(anchor)),  inverse_sigmoid(x) = log(max(clamp(x, 0, 1), eps) / max(1 - clamp(x, 0, 1), eps))
// ($CQ/modules/utils.py:83-87, eps = 1e-5).  The anchors are detached reference windows: only `delta` has a gradient,
// d out / d delta = out (1 - out).  PyTorch runs this as clamp, clamp, rsub, clamp, div, log, add, sigmoid (8 launches
// forward, 3 backward) on a few thousand numbers, seven times per step.
__global__ void __launch_bounds__(256) box_refine_fwd_kernel(const float* __restrict__ delta, const float* __restrict__ anchor,
                                                              long long n, float eps, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = fminf(fmaxf(anchor[i], 0.0f), 1.0f);
  const float x1 = fmaxf(x, eps), x2 = fmaxf(__fsub_rn(1.0f, x), eps);
  const float s = __fadd_rn(delta[i], logf(__fdiv_rn(x1, x2)));
  out[i] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-s)));
}

// ganchor (optional): the chain through inverse_sigmoid as autograd differentiates its clamps -- clamp(a, 0, 1) passes the
// gradient for 0 <= a <= 1, clamp(., min = eps) for values >= eps: ds/da = [0 <= a <= 1] ([x >= eps] / x1 + [1 - x >= eps] / x2)
__global__ void __launch_bounds__(256) box_refine_bwd_kernel(const float* __restrict__ grad, const float* __restrict__ out,
                                                              const float* __restrict__ anchor, long long n, float eps,
                                                              float* __restrict__ gdelta, float* __restrict__ ganchor) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float y = out[i];
  const float gs = __fmul_rn(__fmul_rn(grad[i], __fsub_rn(1.0f, y)), y);   // sigmoid_backward: grad * (1 - y) * y
  gdelta[i] = gs;
  if (ganchor) {
    const float a = anchor[i];
    const float x = fminf(fmaxf(a, 0.0f), 1.0f), r = __fsub_rn(1.0f, x);
    const float d = (x >= eps ? __fdiv_rn(1.0f, fmaxf(x, eps)) : 0.0f) + (r >= eps ? __fdiv_rn(1.0f, fmaxf(r, eps)) : 0.0f);
    ganchor[i] = (a >= 0.0f && a <= 1.0f) ? __fmul_rn(gs, d) : 0.0f;
  }
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_match_cost_f32(const float* logits, const float* boxes, const int64_t* tgt_labels,
                                  const float* tgt_boxes, int p, int b, int q, int c, int g, float w_class, float w_bbox,
                                  float w_giou, float w_rad, float alpha, float gamma, float* cost, void* stream) {
  EFG_CHECK_ARG(p >= 0 && b >= 1 && q >= 0 && c >= 1 && g >= 0 && p % b == 0, "match_cost: bad sizes");
  const long long total = (long long)p * q * g;
  if (total == 0) return EFG_OK;
  EFG_CHECK_ARG(logits && boxes && tgt_labels && tgt_boxes && cost, "match_cost: null pointer");
  hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     boxes, (const long long*)tgt_labels, tgt_boxes, p, b, q, c, g,
                     CostW{w_class, w_bbox, w_giou, w_rad, alpha, gamma}, cost);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_focal_loss_forward_f32(const float* logits, const int32_t* target_class, int layers, int64_t n, int c,
                                          float alpha, float gamma, const float* denom, float* out, void* stream) {
  EFG_CHECK_ARG(layers >= 0 && n >= 0 && c >= 1, "focal_loss: bad sizes");
  if (layers == 0) return EFG_OK;
  EFG_CHECK_ARG(denom && out && (n == 0 || (logits && target_class)), "focal_loss: null pointer");
  // from ~16k elements per layer: split over up to 32 workgroups per layer (a fixed function of the shape: the grouping of
  // the sum, hence its bits, depends on nothing else)
  const long long per_layer = (long long)n * c;
  const int split = (int)std::min<long long>(32, per_layer / 8192);
  unsigned* slots = split >= 2 ? ticket_slots(layers * (1 + split), (hipStream_t)stream) : nullptr;
  if (slots)
    hipLaunchKernelGGL(focal_sum_split_kernel, dim3(layers, split), dim3(1024), 0, (hipStream_t)stream, logits, target_class,
                       (long long)n, c, alpha, gamma, denom, out, slots);
  else
    hipLaunchKernelGGL(focal_sum_kernel, dim3(layers), dim3(1024), 0, (hipStream_t)stream, logits, target_class,
                       (long long)n, c, alpha, gamma, denom, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_focal_loss_backward_f32(const float* logits, const int32_t* target_class, int layers, int64_t n, int c,
                                           float alpha, float gamma, const float* denom, const float* grad_out,
                                           float* grad_logits, void* stream) {
  EFG_CHECK_ARG(layers >= 0 && n >= 0 && c >= 1, "focal_loss: bad sizes");
  const long long total = (long long)layers * n * c;
  if (total == 0) return EFG_OK;
  EFG_CHECK_ARG(logits && target_class && denom && grad_out && grad_logits, "focal_loss: null pointer");
  hipLaunchKernelGGL(focal_grad_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     target_class, (long long)n, c, layers, alpha, gamma, denom, grad_out, grad_logits);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_box_loss_forward_f32(const float* boxes, const float* tgt_boxes, const int64_t* l_idx,
                                        const int64_t* b_idx, const int64_t* q_idx, const int64_t* g_idx, int64_t n,
                                        int layers, int b, int q, int g, const float* denom, float* out, void* stream) {
  EFG_CHECK_ARG(layers >= 0 && n >= 0, "box_loss: bad sizes");
  if (layers == 0) return EFG_OK;
  EFG_CHECK_ARG(denom && out && (n == 0 || (boxes && tgt_boxes && l_idx && b_idx && q_idx && g_idx)), "box_loss: null pointer");
  hipLaunchKernelGGL(box_loss_kernel, dim3(layers), dim3(1024), 0, (hipStream_t)stream, boxes, tgt_boxes,
                     (const long long*)l_idx, (const long long*)b_idx, (const long long*)q_idx, (const long long*)g_idx,
                     (long long)n, b, q, g, denom, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_box_loss_backward_f32(const float* boxes, const float* tgt_boxes, const int64_t* l_idx,
                                         const int64_t* b_idx, const int64_t* q_idx, const int64_t* g_idx, int64_t n,
                                         int layers, int b, int q, int g, const float* denom, const float* grad_out,
                                         float* grad_boxes, void* stream) {
  EFG_CHECK_ARG(layers >= 0 && n >= 0, "box_loss: bad sizes");
  if (n == 0) return EFG_OK;
  EFG_CHECK_ARG(boxes && tgt_boxes && l_idx && b_idx && q_idx && g_idx && denom && grad_out && grad_boxes,
                "box_loss: null pointer");
  hipLaunchKernelGGL(box_loss_grad_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, boxes,
                     tgt_boxes, (const long long*)l_idx, (const long long*)b_idx, (const long long*)q_idx,
                     (const long long*)g_idx, (long long)n, b, q, g, denom, grad_out, grad_boxes);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_box_refine_forward_f32(const float* delta, const float* anchor, int64_t n, float eps, float* out,
                                          void* stream) {
  EFG_CHECK_ARG(n >= 0 && (n == 0 || (delta && anchor && out)), "box_refine: bad arguments");
  if (n == 0) return EFG_OK;
  hipLaunchKernelGGL(box_refine_fwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, delta, anchor,
                     (long long)n, eps, out);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" int efg_box_refine_backward_f32(const float* grad, const float* out, const float* anchor, int64_t n, float eps,
                                           float* grad_delta, float* grad_anchor, void* stream) {
  EFG_CHECK_ARG(n >= 0 && (n == 0 || (grad && out && grad_delta)), "box_refine backward: bad arguments");
  EFG_CHECK_ARG(!grad_anchor || anchor, "box_refine backward: the anchor gradient needs the anchors");
  if (n == 0) return EFG_OK;
  hipLaunchKernelGGL(box_refine_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, grad, out,
                     anchor, (long long)n, eps, grad_delta, grad_anchor);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
