// Rotated BEV overlap / IoU / 3-D IoU and greedy NMS for gfx950.
//
// Replaces efg/operators/src/iou3d_nms/iou3d_nms_kernel.cu (box_overlap :111-239, iou_bev :241-248,
// boxes_overlap_kernel / boxes_iou_bev_kernel :250-268, nms_kernel :270-309, nms_normal_kernel :324-362)
// and the host suppression loop of iou3d_nms.cpp:82-128.  Same per-pair arithmetic (operation order kept so
// that fp32 results only differ through sin/cos/atan2 ULPs); the launch structure is MI355X-native:
//
//   * pair kernels: one lane per (a, b) pair, candidate polygon vertices in LDS columns
//     (pts[k][lane], conflict free) instead of a scratch array, plus an EXACT early reject -- when the
//     centres are further apart than both half-diagonals + margin no vertex can be produced, the reference
//     returns 0 through its cnt == 0 path, and so do we without running the clipper.
//   * nms mask: wave64-native 64 x 64 tiles (one 64-bit word per lane is one tile row), upper triangle only.
//   * nms reduce: stays on the GPU (the reference copies the N*N/64 mask to the host and loops there).
//     One workgroup walks the row blocks; inside a block the 64 x 64 diagonal tile lives in one VGPR and the
//     greedy chain is 64 readlane steps with no memory access; the surviving rows are then OR-ed into the
//     later column blocks by 16 waves with independent (pipelined) loads.
//   * both pair kernels compact the few candidate pairs into an LDS queue so the clipper runs on full waves.
#include "common.h"

namespace efg {
namespace {

constexpr int kMaxPts = 16;  // iou3d_nms_kernel.cu:162 (Point cross_points[16])
constexpr int kPairThreads = 64;
constexpr float kEps = 1e-8f;
constexpr float kMargin = 1e-2f;

struct P2 {
  float x, y;
};

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool rect_cross(P2 p1, P2 p2, P2 q1, P2 q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

struct Box {
  float x, y, dx, dy, cs, sn;  // cs/sn of +heading
  P2 c[4];
};

__device__ __forceinline__ Box load_box(const float* b) {
  Box r;
  r.x = b[0];
  r.y = b[1];
  r.dx = b[3];
  r.dy = b[4];
  r.cs = cosf(b[6]);
  r.sn = sinf(b[6]);
  const float hx = r.dx / 2, hy = r.dy / 2;
  const float xs[4] = {r.x - hx, r.x + hx, r.x + hx, r.x - hx};
  const float ys[4] = {r.y - hy, r.y - hy, r.y + hy, r.y + hy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // rotate_around_center :99-104
    r.c[k].x = (xs[k] - r.x) * r.cs + (ys[k] - r.y) * (-r.sn) + r.x;
    r.c[k].y = (xs[k] - r.x) * r.sn + (ys[k] - r.y) * r.cs + r.y;
  }
  return r;
}

// check_in_box2d :54-64 rotates by -heading: cos(-a) = cs, sin(-a) = -sn (exact symmetries of cosf/sinf).
__device__ __forceinline__ bool in_box(const Box& b, P2 p) {
  const float c = b.cs, s = -b.sn;
  const float rx = (p.x - b.x) * c + (p.y - b.y) * (-s);
  const float ry = (p.x - b.x) * s + (p.y - b.y) * c;
  return fabsf(rx) < b.dx / 2 + kMargin && fabsf(ry) < b.dy / 2 + kMargin;
}

__device__ __forceinline__ bool seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2* ans) {  // :66-97
  if (!rect_cross(p0, p1, q0, q1)) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

// True when the two (margin-inflated) rectangles cannot share a point: every vertex the clipper could emit
// lies inside both inflated rectangles, each of which lies inside the disc of radius half-diagonal +
// sqrt(2) * margin around its centre.  0.05 leaves > 3x slack over that and over fp32 rounding.
__device__ __forceinline__ bool far_apart(const float* a, const float* b) {
  const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]), rb = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
  const float dx = a[0] - b[0], dy = a[1] - b[1], r = ra + rb + 0.05f;
  return dx * dx + dy * dy > r * r;
}

// Polygon-intersection area.  px/py/pa: this lane's LDS columns (stride = blockDim.x floats).
template <int STRIDE>
__device__ __forceinline__ float box_overlap(const float* box_a, const float* box_b, float* px, float* py, float* pa) {
  if (far_apart(box_a, box_b)) return 0.0f;
  const Box A = load_box(box_a), B = load_box(box_b);
  int cnt = 0;
  float cx = 0.f, cy = 0.f;
  auto push = [&](P2 p) {
    if (cnt < kMaxPts) {
      cx += p.x;
      cy += p.y;
      px[cnt * STRIDE] = p.x;
      py[cnt * STRIDE] = p.y;
      ++cnt;
    }
  };
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P2 ans;
      if (seg_intersection(A.c[(i + 1) & 3], A.c[i], B.c[(j + 1) & 3], B.c[j], &ans)) push(ans);
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (in_box(A, B.c[k])) push(B.c[k]);
    if (in_box(B, A.c[k])) push(A.c[k]);
  }
  if (cnt == 0) return 0.0f;
  cx /= cnt;
  cy /= cnt;
  // stable insertion sort by angle == the reference's bubble sort with a strict '>' (:207-215)
  for (int k = 0; k < cnt; ++k) pa[k * STRIDE] = atan2f(py[k * STRIDE] - cy, px[k * STRIDE] - cx);
  for (int k = 1; k < cnt; ++k) {
    const float a = pa[k * STRIDE], x = px[k * STRIDE], y = py[k * STRIDE];
    int m = k - 1;
    while (m >= 0 && pa[m * STRIDE] > a) {
      pa[(m + 1) * STRIDE] = pa[m * STRIDE];
      px[(m + 1) * STRIDE] = px[m * STRIDE];
      py[(m + 1) * STRIDE] = py[m * STRIDE];
      --m;
    }
    pa[(m + 1) * STRIDE] = a;
    px[(m + 1) * STRIDE] = x;
    py[(m + 1) * STRIDE] = y;
  }
  float area = 0.f;
  const float x0 = px[0], y0 = py[0];
  for (int k = 0; k < cnt - 1; ++k) {
    const float ax = px[k * STRIDE] - x0, ay = py[k * STRIDE] - y0;
    const float bx = px[(k + 1) * STRIDE] - x0, by = py[(k + 1) * STRIDE] - y0;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

template <int STRIDE>
__device__ __forceinline__ float iou_bev(const float* a, const float* b, float* px, float* py, float* pa) {
  const float sa = a[3] * a[4], sb = b[3] * b[4], so = box_overlap<STRIDE>(a, b, px, py, pa);
  return so / fmaxf(sa + sb - so, kEps);
}

__device__ __forceinline__ float iou_normal(const float* a, const float* b) {  // :312-322
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float inter = fmaxf(right - left, 0.f) * fmaxf(bottom - top, 0.f);
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, kEps);
}

// Candidate queue shared by both pair kernels.  Rotated boxes rarely overlap (a fraction of a percent of the
// pairs at detection densities), but one candidate lane drags its whole wave through the clipper.  So the
// lanes first run the cheap exact reject, push the surviving (row, col) pairs into an LDS queue with a
// ballot/popcount, and the clipper only runs on full waves of candidates (plus one final partial wave).
constexpr int kQueueCap = 128;  // < 64 pending + 64 pushed per step

struct PairQueue {
  int* rows;
  int* cols;
  int n;  // wave-uniform
  // every lane of the (single-wave) block calls this; returns true when >= 64 entries are pending
  __device__ __forceinline__ bool push(bool cand, int r, int c) {
    const unsigned long long b = __ballot(cand);
    if (cand) {
      const int pos = n + __popcll(b & ((1ULL << lane_id()) - 1ULL));
      rows[pos] = r;
      cols[pos] = c;
    }
    n += __popcll(b);
    return n >= 64;
  }
  // pops up to 64 entries: lane gets (r, c) and returns whether it holds one
  __device__ __forceinline__ bool pop(int* r, int* c) {
    __syncthreads();  // single-wave block: orders the LDS writes above before the reads below
    const int base = n >= 64 ? n - 64 : 0;
    const int e = base + lane_id();
    const bool has = e < n;
    *r = has ? rows[e] : 0;
    *c = has ? cols[e] : 0;
    n = base;
    __syncthreads();
    return has;
  }
};

// mode 0: BEV overlap area, 1: BEV IoU, 2: 3-D IoU (iou3d_nms.py:54-87 fused: height overlap, volumes).
// One single-wave block owns a 64 (rows of a) x 64 (columns of b) tile: lane <-> b column (registers), the a
// row is wave-uniform (scalar loads); rejected pairs store their exact 0 as 256 B coalesced rows.
constexpr int kRowsPerBlock = 64;

__device__ __forceinline__ float pair_value(const float* a, const float* b, int mode, float* px, float* py, float* pa) {
  if (mode == 1) return iou_bev<kPairThreads>(a, b, px, py, pa);
  float v = box_overlap<kPairThreads>(a, b, px, py, pa);
  if (mode == 2) {
    const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2;
    const float b_max = b[2] + b[5] / 2, b_min = b[2] - b[5] / 2;
    const float oh = fmaxf(fminf(a_max, b_max) - fmaxf(a_min, b_min), 0.f);
    const float o3 = v * oh;
    const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
    v = o3 / fmaxf(va + vb - o3, 1e-6f);
  }
  return v;
}

__global__ __launch_bounds__(kPairThreads) void boxes_pair_kernel(const float* __restrict__ boxes_a, int na,
                                                                  const float* __restrict__ boxes_b, int nb, int mode,
                                                                  float* __restrict__ out) {
  __shared__ float pts[3 * kMaxPts * kPairThreads];
  __shared__ int q_rows[kQueueCap], q_cols[kQueueCap];
  float* px = pts + threadIdx.x;
  float* py = px + kMaxPts * kPairThreads;
  float* pa = py + kMaxPts * kPairThreads;
  PairQueue q{q_rows, q_cols, 0};
  const int j = blockIdx.x * kPairThreads + threadIdx.x;
  const int i0 = blockIdx.y * kRowsPerBlock;
  const int jc = j < nb ? j : nb - 1;
  const float bx = boxes_b[(int64_t)jc * 7 + 0], by = boxes_b[(int64_t)jc * 7 + 1];
  const float bw = boxes_b[(int64_t)jc * 7 + 3], bh = boxes_b[(int64_t)jc * 7 + 4];
  const float rb = 0.5f * sqrtf(bw * bw + bh * bh);

  auto drain = [&]() {
    int i, c;
    if (q.pop(&i, &c)) {
      float a[7], b[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        a[k] = boxes_a[(int64_t)i * 7 + k];
        b[k] = boxes_b[(int64_t)c * 7 + k];
      }
      out[(int64_t)i * nb + c] = pair_value(a, b, mode, px, py, pa);
    }
  };

  const int rows = min(na - i0, kRowsPerBlock);
  for (int r = 0; r < rows; ++r) {
    const int i = i0 + r;
    const float ax = boxes_a[(int64_t)i * 7 + 0], ay = boxes_a[(int64_t)i * 7 + 1];
    const float aw = boxes_a[(int64_t)i * 7 + 3], ah = boxes_a[(int64_t)i * 7 + 4];
    const float ra = 0.5f * sqrtf(aw * aw + ah * ah);
    const float dx = ax - bx, dy = ay - by, rr = ra + rb + 0.05f;  // == far_apart(a, b)
    const bool far = dx * dx + dy * dy > rr * rr;
    if (j < nb && far) out[(int64_t)i * nb + j] = 0.0f;
    if (q.push(j < nb && !far, i, j)) drain();
  }
  if (q.n > 0) drain();
}

// mask[i][cb] bit t  <=>  IoU(box i, box cb*64+t) > thresh, for cb >= i/64 and (cb*64+t) > i.
// Tiles below the diagonal are never written nor read.  Lane <-> row of the tile; rotated tiles go through the
// candidate queue and set their bits with ds_or_b64.
// seg (nullable): segment id of every box, ascending; boxes of different segments never suppress each other
// (batched NMS of independent sets in one launch).  Tiles whose segment ranges do not meet write zeros and leave.
template <bool kRotated>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh,
                                                      int col_blocks, unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ seg) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (cb < rb) return;
  __shared__ int seg_c[64];
  int seg_r = 0;
  if (seg) {
    const int row = min(rb * 64 + (int)threadIdx.x, n - 1), col = min(cb * 64 + (int)threadIdx.x, n - 1);
    seg_r = seg[row];
    seg_c[threadIdx.x] = seg[col];
    if (seg[min(rb * 64 + 63, n - 1)] < seg[cb * 64]) {  // sorted ids: the whole tile pairs different segments
      if (rb * 64 + (int)threadIdx.x < n) mask[((int64_t)rb * 64 + threadIdx.x) * col_blocks + cb] = 0ULL;
      return;
    }
  }
  __shared__ float cols[64 * 7];
  __shared__ float rws[64 * 7];
  __shared__ float pts[kRotated ? 3 * kMaxPts * 64 : 1];
  __shared__ int q_rows[kRotated ? kQueueCap : 1], q_cols[kRotated ? kQueueCap : 1];
  __shared__ unsigned long long bits_s[64];
  const int ncol = min(n - cb * 64, 64), nrow = min(n - rb * 64, 64);
  for (int t = threadIdx.x; t < ncol * 7; t += 64) cols[t] = boxes[(int64_t)cb * 64 * 7 + t];
  for (int t = threadIdx.x; t < nrow * 7; t += 64) rws[t] = boxes[(int64_t)rb * 64 * 7 + t];
  bits_s[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x;
  const int lr = lane < nrow ? lane : nrow - 1;
  const float* a = rws + lr * 7;
  if constexpr (!kRotated) {
    unsigned long long bits = 0;
    const int start = (rb == cb) ? lane + 1 : 0;
    for (int t = start; t < ncol; ++t)
      if ((!seg || seg_c[t] == seg_r) && iou_normal(a, cols + t * 7) > thresh) bits |= 1ULL << t;
    if (lane < nrow) mask[((int64_t)rb * 64 + lane) * col_blocks + cb] = bits;
    return;
  } else {
    float* px = pts + lane;
    float* py = px + kMaxPts * 64;
    float* pa = py + kMaxPts * 64;
    PairQueue q{q_rows, q_cols, 0};
    auto drain = [&]() {
      int r, c;
      if (q.pop(&r, &c)) {
        if (iou_bev<64>(rws + r * 7, cols + c * 7, px, py, pa) > thresh) atomicOr(&bits_s[r], 1ULL << c);
      }
    };
    const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]);
    for (int t = 0; t < ncol; ++t) {
      const float* b = cols + t * 7;
      const float rbb = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
      const float dx = a[0] - b[0], dy = a[1] - b[1], rr = ra + rbb + 0.05f;  // == far_apart(a, b)
      const bool cand = lane < nrow && (rb != cb || t > lane) && !(dx * dx + dy * dy > rr * rr) &&
                        (!seg || seg_c[t] == seg_r);
      if (q.push(cand, lane, t)) drain();
    }
    if (q.n > 0) drain();
    __syncthreads();
    if (lane < nrow) mask[((int64_t)rb * 64 + lane) * col_blocks + cb] = bits_s[lane];
  }
}

// One 1024-thread block; remv[] (one word per column block) lives in LDS.  The greedy chain is inherently
// serial over the boxes, so the kernel is built around its critical path:
//   * wave 0 is the chain wave.  For row block rb it holds two 64 x 64 tiles in VGPRs (lane <-> row): the
//     diagonal tile (column block rb) and the tile of column block rb+1, both prefetched one block ahead.
//     The chain is 64 scalar steps (v_readlane into SGPRs, no memory access); the suppression the kept rows
//     cast on the NEXT block is an in-wave OR of the second tile ("carry"), so nothing on the path from one
//     block's chain to the next touches memory.
//   * waves 1..15 push, one block behind, the kept rows of block rb-1 into the column blocks >= rb+1 (lanes
//     run over column blocks: coalesced, independent, branch-free loads; partial words meet in LDS with
//     ds_or_b64) while wave 0 is already chaining block rb.  One barrier per block.
constexpr int kReduceThreads = 1024;

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kReduceThreads) void nms_reduce_kernel(const unsigned long long* __restrict__ mask, int n,
                                                                    int col_blocks, int64_t* __restrict__ keep,
                                                                    int* __restrict__ num_keep) {
  extern __shared__ unsigned long long remv[];  // [col_blocks] + 2 words: kept sets of the last two blocks
  unsigned long long* kept_s = remv + col_blocks;
  // readfirstlane makes the wave id provably uniform: `if (wave == 0)` becomes a scalar branch and the chain
  // state (cur / kept / carry) stays in SGPRs instead of being dragged into VGPRs by a "divergent" branch
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int t = threadIdx.x; t < col_blocks; t += kReduceThreads) remv[t] = 0;
  __syncthreads();

  auto tile = [&](int rb, int cb) -> unsigned long long {  // lane's word of tile (rb, cb), 0 outside
    const int64_t row = (int64_t)rb * 64 + lane;
    return (rb < col_blocks && cb < col_blocks && row < n) ? mask[row * col_blocks + cb] : 0ULL;
  };

  int kept_total = 0;
  unsigned long long carry = 0;  // wave 0: what block rb-1's kept rows suppress in block rb
  unsigned long long diag_c = 0, next_c = 0;
  if (wave == 0) {
    diag_c = tile(0, 0);
    next_c = tile(0, 1);
  }
  for (int rb = 0; rb < col_blocks; ++rb) {
    if (wave == 0) {
      const unsigned long long diag_n = tile(rb + 1, rb + 1), next_n = tile(rb + 1, rb + 2);  // prefetch
      const int rows = min(n - rb * 64, 64);
      unsigned long long cur = uniform64(remv[rb]) | carry;
      unsigned long long kept = 0;
      const int dlo = (int)(unsigned)diag_c, dhi = (int)(unsigned)(diag_c >> 32);
      // visit only the rows that are still alive: i = lowest clear bit of cur at or above the cursor
      const unsigned long long valid = rows == 64 ? ~0ULL : ((1ULL << rows) - 1ULL);
      unsigned long long todo = ~cur & valid;
      while (todo) {
        const int i = __builtin_ctzll(todo);
        kept |= 1ULL << i;
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(dlo, i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane(dhi, i);
        cur |= (((unsigned long long)hi << 32) | lo) | (1ULL << i);
        todo = ~cur & valid;  // diag bits are all above i, so everything at or below i is now set in cur
      }
      if ((kept >> lane) & 1ULL)
        keep[kept_total + __popcll(kept & ((1ULL << lane) - 1ULL))] = (int64_t)rb * 64 + lane;
      kept_total += __popcll(kept);
      if (lane == 0) kept_s[rb & 1] = kept;
      unsigned long long c = ((kept >> lane) & 1ULL) ? next_c : 0ULL;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) c |= __shfl_xor(c, d, 64);
      carry = uniform64(c);
      diag_c = diag_n;
      next_c = next_n;
    } else if (rb >= 1) {
      const unsigned long long kept = kept_s[(rb - 1) & 1];  // block rb-1 is always a full block
      const int64_t row0 = (int64_t)(rb - 1) * 64;
      for (int cb = rb + 1 + lane; cb < col_blocks; cb += 64) {
        unsigned long long acc = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const int i = (wave - 1) + 15 * r;  // rows 0..63 over the 15 push waves
          const int ic = i < 64 ? i : 63;
          const unsigned long long w = mask[(row0 + ic) * col_blocks + cb];
          acc |= (i < 64 && ((kept >> ic) & 1ULL)) ? w : 0ULL;
        }
        if (acc) atomicOr(&remv[cb], acc);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_keep = kept_total;
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_boxes_bev_f32(const float* boxes_a, int na, const float* boxes_b, int nb, int mode, float* out,
                                 void* stream) {
  EFG_CHECK_ARG(na >= 0 && nb >= 0, "efg_boxes_bev_f32: negative box count (%d, %d)", na, nb);
  EFG_CHECK_ARG(mode >= 0 && mode <= 2, "efg_boxes_bev_f32: mode must be 0 (overlap) | 1 (iou) | 2 (iou3d), got %d",
                mode);
  if (na == 0 || nb == 0) return EFG_OK;
  EFG_CHECK_ARG(boxes_a && boxes_b && out, "efg_boxes_bev_f32: null pointer");
  const int64_t gy = ceil_div(na, kRowsPerBlock);
  EFG_CHECK_ARG(gy <= 65535 * 16, "efg_boxes_bev_f32: na = %d too large", na);
  // gridDim.y is limited to 65535: chunk the rows
  for (int64_t y0 = 0; y0 < gy; y0 += 65535) {
    const int rows0 = (int)(y0 * kRowsPerBlock);
    const int ny = (int)((gy - y0) < 65535 ? (gy - y0) : 65535);
    const int na_chunk = (na - rows0) < ny * kRowsPerBlock ? (na - rows0) : ny * kRowsPerBlock;
    hipLaunchKernelGGL(boxes_pair_kernel, dim3((unsigned)ceil_div(nb, kPairThreads), (unsigned)ny), dim3(kPairThreads),
                       0, (hipStream_t)stream, boxes_a + (int64_t)rows0 * 7, na_chunk, boxes_b, nb, mode,
                       out + (int64_t)rows0 * nb);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

extern "C" size_t efg_nms_workspace_bytes(int n) {
  const size_t cb = (size_t)ceil_div(n > 0 ? n : 1, 64);
  return align_up((size_t)(n > 0 ? n : 1) * cb * sizeof(unsigned long long), 256);
}

static int nms_run(const float* boxes_sorted, const int* seg, int n, float thresh, int rotated, int64_t* keep,
                   int* num_keep, void* ws, size_t ws_bytes, void* stream);

extern "C" int efg_nms_f32(const float* boxes_sorted, int n, float thresh, int rotated, int64_t* keep, int* num_keep,
                           void* ws, size_t ws_bytes, void* stream) {
  return nms_run(boxes_sorted, nullptr, n, thresh, rotated, keep, num_keep, ws, ws_bytes, stream);
}

extern "C" int efg_nms_segmented_f32(const float* boxes_sorted, const int32_t* segment, int n, float thresh, int rotated,
                                     int64_t* keep, int* num_keep, void* ws, size_t ws_bytes, void* stream) {
  EFG_CHECK_ARG(segment || n == 0, "efg_nms_segmented_f32: segment ids are null");
  return nms_run(boxes_sorted, segment, n, thresh, rotated, keep, num_keep, ws, ws_bytes, stream);
}

static int nms_run(const float* boxes_sorted, const int* seg, int n, float thresh, int rotated, int64_t* keep,
                   int* num_keep, void* ws, size_t ws_bytes, void* stream) {
  EFG_CHECK_ARG(n >= 0, "efg_nms_f32: negative box count %d", n);
  EFG_CHECK_ARG(num_keep, "efg_nms_f32: num_keep is null");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    EFG_HIP_TRY(hipMemsetAsync(num_keep, 0, sizeof(int), st));
    return EFG_OK;
  }
  EFG_CHECK_ARG(boxes_sorted && keep && ws, "efg_nms_f32: null pointer");
  EFG_CHECK_ARG(ws_bytes >= efg_nms_workspace_bytes(n), "efg_nms_f32: workspace too small (%zu < %zu)", ws_bytes,
                efg_nms_workspace_bytes(n));
  const int col_blocks = (int)ceil_div(n, 64);
  EFG_CHECK_ARG(col_blocks <= 8000, "efg_nms_f32: n = %d exceeds the 512000-box limit of the on-chip reduce", n);
  auto* mask = static_cast<unsigned long long*>(ws);
  if (rotated)
    hipLaunchKernelGGL((nms_mask_kernel<true>), dim3(col_blocks, col_blocks), dim3(64), 0, st, boxes_sorted, n, thresh,
                       col_blocks, mask, seg);
  else
    hipLaunchKernelGGL((nms_mask_kernel<false>), dim3(col_blocks, col_blocks), dim3(64), 0, st, boxes_sorted, n, thresh,
                       col_blocks, mask, seg);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(kReduceThreads),
                     (size_t)(col_blocks + 2) * sizeof(unsigned long long), st, mask, n, col_blocks, keep, num_keep);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
