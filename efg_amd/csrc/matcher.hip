// Linear sum assignment on the GPU (SURVEY.md section 8(f) "GPU matcher").
//
// The reference moves every cost matrix to the host and calls scipy.optimize.linear_sum_assignment
// ($CQ/modules/matcher.py:86-91): one device->host transfer + a stream drain per call, in the middle of the
// training step.  This kernel keeps the matching on the device so that the step has no host round trip at all.
//
// Algorithm: the rectangular shortest-augmenting-path method of Crouse (IEEE TAES 2016) that scipy implements,
// in fp64 with scipy's operation order AND scipy's tie-breaking, so the assignment is identical to scipy's for
// every input, ties included (oracle/efg_oracle.c:oracle_lsap is the pinned CPU twin).  scipy scans the
// not-yet-visited columns in the order of its `remaining` list (initially reversed, compacted by moving the
// last entry into the freed slot) and keeps the first lowest entry unless a later one is unassigned; that is a
// lexicographic minimum over (value, assigned?, +-position), which is what the block-wide reduction computes.
//
// One workgroup per problem (problems are tiny: <= a few thousand queries x <= a few hundred boxes; all
// (layer, scene) problems of a step run side by side).  Column state lives in LDS; a column scan is one
// coalesced pass over the threads, two barriers per augmenting-path step.
#include "common.h"

#include <math.h>

namespace efg {
namespace {

constexpr int kLsapThreads = 256;

struct Cand {
  double val;
  int key;  // (assigned ? 1 << 30 : 0) + (assigned ? pos : nc - pos): smaller wins
  int col;
};

__device__ __forceinline__ bool better(const Cand& a, const Cand& b) {
  return a.val < b.val || (a.val == b.val && a.key < b.key);
}

__device__ __forceinline__ Cand wave_min(Cand c) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    Cand o;
    o.val = __shfl_xor(c.val, d, 64);
    o.key = __shfl_xor(c.key, d, 64);
    o.col = __shfl_xor(c.col, d, 64);
    if (better(o, c)) c = o;
  }
  return c;
}

// LDS layout helper (dynamic shared memory, 8-byte aligned first)
struct Lds {
  double* spc;     // [nc] shortest path cost
  double* v;       // [nc] column duals
  double* u;       // [nr] row duals
  int* path;       // [nc]
  int* row4col;    // [nc]
  int* pos;        // [nc] position in `remaining`, -1 once visited (== scipy's SC)
  int* remaining;  // [nc]
  int* col4row;    // [nr]
  int* sr;         // [nr] visited rows of the current augmentation (== scipy's SR, as a list)
};

__host__ __device__ inline size_t lds_bytes(int nr, int nc) {
  return sizeof(double) * (2 * (size_t)nc + nr) + sizeof(int) * (4 * (size_t)nc + 2 * (size_t)nr);
}

__global__ __launch_bounds__(kLsapThreads) void lsap_kernel(const float* __restrict__ cost, int nq, int g_stride,
                                                            const int* __restrict__ ng_dev,
                                                            int64_t* __restrict__ query_of_gt,
                                                            int* __restrict__ status) {
  extern __shared__ double lds_raw[];
  __shared__ Cand wave_best[kLsapThreads / 64];
  __shared__ int s_i, s_sink, s_num_remaining, s_nsr, s_fail;
  __shared__ double s_min_val;

  const int p = blockIdx.x, tid = threadIdx.x;
  const int ng = ng_dev[p];
  const float* c = cost + (size_t)p * nq * g_stride;
  int64_t* out = query_of_gt + (size_t)p * g_stride;
  for (int g = tid; g < g_stride; g += kLsapThreads) out[g] = -1;
  if (tid == 0 && status) status[p] = 0;
  if (ng <= 0 || nq <= 0) return;
  const bool transposed = nq > ng;  // scipy transposes so that rows are the smaller side
  const int nr = transposed ? ng : nq, nc = transposed ? nq : ng;
  // cost(i, j) of the (possibly transposed) problem
  const size_t rs = transposed ? 1 : (size_t)g_stride, cs = transposed ? (size_t)g_stride : 1;

  Lds L;
  L.spc = lds_raw;
  L.v = L.spc + nc;
  L.u = L.v + nc;
  L.path = reinterpret_cast<int*>(L.u + nr);
  L.row4col = L.path + nc;
  L.pos = L.row4col + nc;
  L.remaining = L.pos + nc;
  L.col4row = L.remaining + nc;
  L.sr = L.col4row + nr;

  for (int j = tid; j < nc; j += kLsapThreads) {
    L.v[j] = 0.0;
    L.row4col[j] = -1;
  }
  for (int i = tid; i < nr; i += kLsapThreads) {
    L.u[i] = 0.0;
    L.col4row[i] = -1;
  }
  if (tid == 0) s_fail = 0;
  __syncthreads();

  for (int cur = 0; cur < nr; ++cur) {
    for (int j = tid; j < nc; j += kLsapThreads) {
      L.spc[j] = INFINITY;
      L.pos[j] = nc - 1 - j;  // remaining[it] = nc - it - 1
      L.remaining[j] = nc - 1 - j;
    }
    if (tid == 0) {
      s_i = cur;
      s_sink = -1;
      s_num_remaining = nc;
      s_nsr = 0;
      s_min_val = 0.0;
    }
    __syncthreads();

    while (true) {
      const int i = s_i;
      const double min_val = s_min_val, ui = L.u[i];
      Cand best{INFINITY, 0x7fffffff, -1};
      for (int j = tid; j < nc; j += kLsapThreads) {
        const int pj = L.pos[j];
        if (pj < 0) continue;
        const double r = min_val + (double)c[(size_t)i * rs + (size_t)j * cs] - ui - L.v[j];
        double s = L.spc[j];
        if (r < s) {
          L.path[j] = i;
          L.spc[j] = s = r;
        }
        const bool assigned = L.row4col[j] != -1;
        const Cand cand{s, assigned ? (1 << 30) + pj : nc - pj, j};
        if (better(cand, best)) best = cand;
      }
      best = wave_min(best);
      if ((tid & 63) == 0) wave_best[tid >> 6] = best;
      __syncthreads();
      if (tid == 0) {
        Cand b = wave_best[0];
#pragma unroll
        for (int w = 1; w < kLsapThreads / 64; ++w)
          if (better(wave_best[w], b)) b = wave_best[w];
        L.sr[s_nsr++] = i;  // SR[i] = true
        if (b.col < 0 || !(b.val < INFINITY)) {
          s_fail = 1;  // infeasible / NaN costs (scipy raises ValueError)
        } else {
          const int j = b.col;
          s_min_val = b.val;
          if (L.row4col[j] == -1) s_sink = j; else s_i = L.row4col[j];
          // SC[j] = true; remaining[index] = remaining[--num_remaining]
          const int idx = L.pos[j], last = L.remaining[--s_num_remaining];
          L.remaining[idx] = last;
          L.pos[last] = idx;
          L.pos[j] = -1;
        }
      }
      __syncthreads();
      if (s_fail || s_sink != -1) break;
    }
    if (s_fail) break;

    // dual updates
    const double min_val = s_min_val;
    const int nsr = s_nsr, sink = s_sink;
    for (int t = tid; t < nsr; t += kLsapThreads) {
      const int r = L.sr[t];
      if (r == cur) L.u[r] += min_val; else L.u[r] += min_val - L.spc[L.col4row[r]];
    }
    __syncthreads();  // u reads col4row / spc before they change below
    for (int j = tid; j < nc; j += kLsapThreads)
      if (L.pos[j] < 0) L.v[j] -= min_val - L.spc[j];
    if (tid == 0) {  // augment along the path
      int j = sink;
      while (true) {
        const int r = L.path[j];
        L.row4col[j] = r;
        const int t = L.col4row[r];
        L.col4row[r] = j;
        j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }

  if (s_fail) {
    if (tid == 0 && status) status[p] = 1;
    return;
  }
  if (transposed) {
    for (int g = tid; g < nr; g += kLsapThreads) out[g] = L.col4row[g];
  } else {
    for (int q = tid; q < nr; q += kLsapThreads) out[L.col4row[q]] = q;
  }
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_lsap_f32(const float* cost, int n_problems, int nq, int g_stride, const int32_t* ng,
                            int64_t* query_of_gt, int32_t* status, void* stream) {
  EFG_CHECK_ARG(n_problems >= 0 && nq >= 0 && g_stride >= 0, "efg_lsap_f32: negative size (%d, %d, %d)", n_problems,
                nq, g_stride);
  if (n_problems == 0 || g_stride == 0) return EFG_OK;
  EFG_CHECK_ARG(ng && query_of_gt, "efg_lsap_f32: null pointer");
  EFG_CHECK_ARG(cost || nq == 0, "efg_lsap_f32: cost is null");
  // rows = min side, cols = max side; bound both by what fits the 160 KB LDS of a gfx950 CU
  const int big = nq > g_stride ? nq : g_stride, small = nq > g_stride ? g_stride : nq;
  const size_t lds = lds_bytes(small, big);
  EFG_CHECK_ARG(lds <= 150 * 1024, "efg_lsap_f32: problem %d x %d needs %zu B of LDS (limit 153600)", nq, g_stride,
                lds);
  EFG_ALLOW_DYNAMIC_LDS(lsap_kernel, 150 * 1024);
  hipLaunchKernelGGL(lsap_kernel, dim3(n_problems), dim3(kLsapThreads), lds, (hipStream_t)stream, cost, nq, g_stride,
                     ng, query_of_gt, status);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}
