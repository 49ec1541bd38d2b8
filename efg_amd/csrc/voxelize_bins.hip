// Hard voxelization, generation 2: points binned into BEV supercells, per-bin cell tables in LDS.
//
// Replaces efg::hard_voxelize (efg/operators/src/voxelize/voxelization.h:51-69; CPU semantics
// voxelization_cpu.cpp:43-99: voxels in first-occurrence order, at most max_points points per voxel in point order,
// processing stops at the first point that would open voxel number max_voxels).  Bit-exact.
//
// The cloud is shuffled (PointShuffle p=1.0), so any per-cell state kept in HBM costs a random device-scope access per
// point and visit (generation 1, voxelize_hash.hip: 8.7x the algorithmic bytes).  Here the only random HBM accesses
// are ONE 8-byte + ONE 32-byte scattered store per point (the binning); every per-cell decision is taken in LDS:
//
//   K1 bin_count   a workgroup stages a 1024-point tile in LDS with 16-byte coalesced loads, computes each point's
//                  supercell (8 x 8 x <=40 cells), aggregates the tile's points per supercell in an LDS hash table and
//                  reserves their places with ONE returning atomic per (tile, supercell) -- on a counter PRIVATE TO
//                  THE XCD the workgroup runs on (count[xcd][supercell]), so the atomic is executed in that XCD's L2.
//                  (A device-scope atomic is executed past the L2s, at the memory side: the ~90k group atomics of a
//                  2 x 180k-point call took 35 us of a 44 us kernel, `scripts/vox_timeline.py`.)
//   K2 bin_scan    exclusive scan over the count[xcd][supercell] array (1024 supercells per workgroup, chunk totals
//                  chained by decoupled look-back): the place of every (supercell, xcd) group -- a supercell's eight
//                  groups are adjacent, so a bin is one contiguous range -- plus the two work lists of K4 / K5
//                  (bins of <= 256 points, bins above).
//   K3 bin_scatter writes {point index, cell inside the supercell} and the padded row of every point to its place.
//   K4 first       per bin: first occurrence of every cell = minimum point index (ds_min in LDS; a wave per bin with
//                  an LDS hash table for bins of <= 256 points, a workgroup with the dense 2560-cell table above);
//                  the winners set their bit in a per-scene bitmap over the points.  The last workgroup of a scene
//                  scans the bitmap: prefix[word] = number of first points before it = the voxel ids of the serial
//                  loop; rank == max_voxels marks i_break (the reference's `break`).
//   K5 write       per bin again: points before i_break are counting-sorted by cell in LDS (count, scan, scatter), the
//                  occupied cells are compacted into a voxel list, then one thread per voxel takes the max_points
//                  lowest point indices in order, looks its voxel id up in the bitmap prefix (rank of its first point)
//                  and writes rows, zero padding, coordinates, count and the fused mean.
// 6 kernels + 1 clear kernel (counters, chunk totals, look-back records) per call.  All scenes of a batch go through one launch per
// stage (blockIdx.y = scene).  K4 / K5 run a fixed grid whose workgroups loop over the work lists (a launch sized for
// the worst case -- 19k workgroups, 5 % of them with work -- spent 20 us on dispatching the empty ones).
#include "voxelize_common.h"

#include <map>
#include <mutex>
#include <utility>

namespace efg {
namespace {

constexpr int kTile = 1024;            // points per workgroup in K1 / K3 (256 threads x 4)
constexpr int kSB = 3;                 // supercell = 8 x 8 x min(grid_z, 40) cells (one BEV token column of the
constexpr int kSXY = 1 << kSB;         // res18 backbone's stride-8 map): the synthetic Waymo scenes put 80 % of their
constexpr int kSZ = 40;                // points into ~5 % of the columns, smaller bins = more workgroups on them
constexpr int kCells = kSXY * kSXY * kSZ;  // 2560 cells: the dense table of a big bin = 10 KB of LDS
constexpr int kSegLds = 1024;          // a big bin's cell-sorted point list lives in LDS up to this many points
constexpr int kSmall = 256;            // bins up to this many points are handled by one wave
constexpr int kSmallT = 512;           // hash slots of a wave's bin
constexpr int kAggSlots = 2048;        // LDS hash slots of the per-tile aggregation in K1
constexpr int kXcd = 8;                // XCDs of an MI355X: one private counter per supercell and XCD
constexpr int kScanChunk = 1024;       // supercells per workgroup of K2 (256 threads x 4, eight counters each)
constexpr int kSplit2 = 768;           // K5: a big bin above this many points is written by 2 workgroups (cells by x parity),
constexpr int kSplit4 = 1536;          // above this by 4 (x and y parity): the heaviest bins set the length of the launch
constexpr int kItemGrid = 1024;        // workgroups of K4 / K5 over all scenes (each loops over the bins)
constexpr unsigned kNone = 0xffffffffu;

// Optional per-workgroup timeline (efg_hard_voxelize_debug_timeline): thread 0 of every workgroup stores the 100 MHz
// wall clock at up to 8 markers per kernel.  nullptr in normal operation (one predictable branch per marker).
constexpr int kDbgMaxWg = 1 << 15;
__device__ __forceinline__ void mark(unsigned long long* dbg, int kernel, int marker) {
  if (dbg != nullptr && threadIdx.x == 0) {
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < (unsigned)kDbgMaxWg) dbg[(size_t)(kernel * 8 + marker) * kDbgMaxWg + wg] = wall_clock64();
  }
}
unsigned long long* g_dbg = nullptr;

struct BinGeom {
  int sz, nsx, nsy, nsz, s_scene, cells;
  int s_stride;  // s_scene rounded up to 4 supercells: a scene's counters are whole 32-counter thread chunks of K2
};

struct SceneWords {
  int wb[kMaxBatch + 1];  // first bitmap word of every scene
};

BinGeom bin_geom(const VoxGeom& g) {
  BinGeom b;
  b.sz = std::min(g.grid[2], kSZ);
  b.nsx = (g.grid[0] + kSXY - 1) / kSXY;
  b.nsy = (g.grid[1] + kSXY - 1) / kSXY;
  b.nsz = (g.grid[2] + b.sz - 1) / b.sz;
  const long long s = (long long)b.nsx * b.nsy * b.nsz;
  b.s_scene = s < (1ll << 26) ? (int)s : -1;
  b.cells = kSXY * kSXY * b.sz;
  b.s_stride = b.s_scene < 0 ? -1 : (b.s_scene + 3) / 4 * 4;
  return b;
}

__device__ __forceinline__ void bin_of(int cx, int cy, int cz, const BinGeom& bg, unsigned& sc, unsigned& lc) {
  const int bz = cz / bg.sz;
  sc = (unsigned)((bz * bg.nsy + (cy >> kSB)) * bg.nsx + (cx >> kSB));
  lc = (unsigned)((((cz - bz * bg.sz) << kSB) + (cy & (kSXY - 1))) << kSB) + (cx & (kSXY - 1));
}

// the XCD this wave runs on (hardware register XCC_ID, bits 3:0)
__device__ __forceinline__ unsigned xcd_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4): immediate = (size - 1) << 11 | offset << 6 | id
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (kXcd - 1);
}

// Copy rows [row0, row0 + nrows) of the [total_rows, f] matrix into LDS as one flat float stream with 16-byte loads
// (a 20 / 24-byte row read by its own lane costs three strided 4-byte loads per row).  Returns the LDS float index
// of row 0's first element.
__device__ __forceinline__ int stage_rows(const float* __restrict__ pts, long long row0, int nrows, int f,
                                          long long total_rows, float* lds) {
  const long long s = row0 * f, e = s + (long long)nrows * f, s4 = s & ~3ll, lim = total_rows * f;
  const bool aligned = (reinterpret_cast<uintptr_t>(pts) & 15) == 0;
  for (long long k = s4 + threadIdx.x * 4; k < e; k += 256 * 4) {
    float4 v;
    if (aligned && k + 3 < lim) {
      v = *reinterpret_cast<const float4*>(pts + k);
    } else {
      v.x = k < lim ? pts[k] : 0.0f;
      v.y = k + 1 < lim ? pts[k + 1] : 0.0f;
      v.z = k + 2 < lim ? pts[k + 2] : 0.0f;
      v.w = k + 3 < lim ? pts[k + 3] : 0.0f;
    }
    *reinterpret_cast<float4*>(lds + (k - s4)) = v;
  }
  return (int)(s - s4);
}

// K1 ------------------------------------------------------------------------------------------------------------
// pos[i] = xcd << 28 | place of point i inside its (supercell, xcd) group; kNone for points outside the range.
// counters, chunk totals and look-back records of a call: cleared words
__global__ void __launch_bounds__(256) vox_clear_kernel(unsigned* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n) p[i + u * 256] = 0u;
  }
}

template <bool STAGE>
__global__ void __launch_bounds__(256)
vox_bin_count_kernel(const float* __restrict__ pts, SceneOffsets so, SceneWords sw, int f, VoxGeom g, BinGeom bg,
                     unsigned* __restrict__ count, unsigned* __restrict__ pos, unsigned char* __restrict__ flags,
                     unsigned* __restrict__ records, int record_words, unsigned long long* dbg) {
  // the staged tile: kTile * f + 8 floats of dynamic LDS (sized by the launch for THIS row width: 20.5 KB at 5 features
  // instead of the 32.8 KB of the widest staged row -- with the 16 KB of the aggregation table four workgroups per CU instead of three)
  extern __shared__ __attribute__((aligned(16))) float stage[];
  __shared__ unsigned hkey[kAggSlots], hcnt[kAggSlots];
  mark(dbg, 0, 0);
  const int scene = blockIdx.y, tid = threadIdx.x;
  // library-owned state (see VoxState): the look-back records of the two scans and the error word of THIS call, all used by
  // later launches only, are zeroed here -- the counters are left zero by K2 of the previous call
  if (records && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = tid; i < record_words; i += 256) records[i] = 0u;
  const long long beg = so.off[scene], end = so.off[scene + 1];
  const long long row0 = beg + (long long)blockIdx.x * kTile;
  const int nrows = (int)max(0ll, min((long long)kTile, end - row0));
  const unsigned xcd = xcd_id();
  // [xcd][supercell]: an XCD's counters are its own contiguous range -- no cache line is shared between two XCDs' L2s
  unsigned* cnt_x = count + ((size_t)scene * kXcd + xcd) * bg.s_stride;
  if (tid < kTile / 16) {  // this tile's first-point flags (one byte per point) start empty
    const long long fb = (long long)sw.wb[scene] * 32 + (long long)blockIdx.x * kTile + tid * 16;
    if (fb < (long long)sw.wb[scene + 1] * 32) *reinterpret_cast<uint4*>(flags + fb) = make_uint4(0, 0, 0, 0);
  }
  if (nrows == 0) return;
  int shift = 0;
  if (STAGE) shift = stage_rows(pts, row0, nrows, f, so.off[kMaxBatch], stage);
  for (int s = tid; s < kAggSlots; s += 256) {
    hkey[s] = kNone;
    hcnt[s] = 0u;
  }
  __syncthreads();
  mark(dbg, 0, 1);   // tile staged
  unsigned slot[4], rank[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = tid + 256 * j;
    slot[j] = kNone;
    rank[j] = 0u;
    if (p < nrows) {
      const float* r = STAGE ? stage + shift + p * f : pts + (row0 + p) * f;
      int cx, cy, cz;
      if (cell_of(r[0], r[1], r[2], g, cx, cy, cz)) {
        unsigned sc, lc;
        bin_of(cx, cy, cz, bg, sc, lc);
        unsigned h = (sc * 2654435761u) >> 21;
        while (true) {
          const unsigned k = atomicCAS(&hkey[h], kNone, sc);
          if (k == kNone || k == sc) break;
          h = (h + 1) & (kAggSlots - 1);
        }
        rank[j] = atomicAdd(&hcnt[h], 1u);
        slot[j] = h;
      }
    }
  }
  __syncthreads();
  mark(dbg, 0, 2);   // LDS aggregation done
  {  // group size -> first place of the group inside its (supercell, xcd) range.  The counter belongs to this XCD alone, so
     // the read-modify-write may stay in this XCD's L2 (workgroup scope: no sc1, not forwarded to the memory side); the
     // L2 is written back at the end of the kernel like any other store.  All of a thread's atomics are in flight
     // together.
    unsigned r[kAggSlots / 256];
#pragma unroll
    for (int i = 0; i < kAggSlots / 256; ++i) {
      const int sl = tid + 256 * i;
      r[i] = hkey[sl] != kNone ? __hip_atomic_fetch_add(cnt_x + hkey[sl], hcnt[sl], __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_WORKGROUP)
                               : 0u;
    }
#pragma unroll
    for (int i = 0; i < kAggSlots / 256; ++i) hcnt[tid + 256 * i] = r[i];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = tid + 256 * j;
    if (p < nrows) pos[row0 + p] = slot[j] == kNone ? kNone : (xcd << 28) | (hcnt[slot[j]] + rank[j]);
  }
  mark(dbg, 0, 3);   // group atomics returned, places written
}

// K2 ------------------------------------------------------------------------------------------------------------
// part[chunk] = 1 << 63 | chunk's points << 30 | big bins << 15 | small bins; 0 = not published yet.
__global__ void __launch_bounds__(256)
vox_bin_scan_kernel(SceneOffsets so, BinGeom bg, unsigned* __restrict__ count, unsigned long long* __restrict__ part,
                    int nchunks, unsigned* __restrict__ basex, uint4* __restrict__ small_list,
                    uint4* __restrict__ big_list, unsigned* __restrict__ nlist, unsigned* __restrict__ err,
                    int self_clean, unsigned long long* dbg) {
  __shared__ int smem[17];
  __shared__ unsigned long long s_w[4];
  mark(dbg, 1, 0);
  const int scene = blockIdx.y, tid = threadIdx.x, chunk = blockIdx.x;
  const size_t xbase = (size_t)scene * bg.s_stride * kXcd;
  const int sc0 = chunk * kScanChunk + tid * 4;           // this thread's four supercells
  uint4 v[kXcd];                                          // v[x] = counters of XCD x for supercells sc0 .. sc0 + 3
  int sum = 0, ns = 0, nb = 0;
  unsigned tot[4];
#pragma unroll
  for (int x = 0; x < kXcd; ++x)
    v[x] = sc0 < bg.s_stride ? *reinterpret_cast<const uint4*>(count + xbase + (size_t)x * bg.s_stride + sc0)
                             : make_uint4(0, 0, 0, 0);
  // library-owned counters (see bins_hard_voxelize): every counter is read exactly once, here -- and left zero for the next call
  if (self_clean && sc0 < bg.s_stride) {
#pragma unroll
    for (int x = 0; x < kXcd; ++x) *reinterpret_cast<uint4*>(count + xbase + (size_t)x * bg.s_stride + sc0) = make_uint4(0, 0, 0, 0);
  }
  tot[0] = tot[1] = tot[2] = tot[3] = 0u;
#pragma unroll
  for (int x = 0; x < kXcd; ++x) {
    tot[0] += v[x].x;
    tot[1] += v[x].y;
    tot[2] += v[x].z;
    tot[3] += v[x].w;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    sum += (int)tot[b];
    ns += (tot[b] > 0u && tot[b] <= (unsigned)kSmall) ? 1 : 0;
    nb += tot[b] > (unsigned)kSplit4 ? 4 : tot[b] > (unsigned)kSplit2 ? 2 : tot[b] > (unsigned)kSmall ? 1 : 0;   // big ITEMS
  }
  int csum, clist;
  const int psum = block_exclusive_scan(sum, smem, &csum);
  const int plist = block_exclusive_scan(ns | (nb << 15), smem, &clist);   // 1024 supercells (<= 4096 items) per chunk: 15 bits each
  unsigned long long* part_s = part + (size_t)scene * nchunks;
  if (tid == 0)  // publish this chunk's totals at once; nothing has been waited for
    __hip_atomic_store(part_s + chunk, (1ull << 63) | ((unsigned long long)(unsigned)csum << 30) | (unsigned)clist,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  mark(dbg, 1, 1);   // chunk reduced + published
  // Decoupled look-back: the totals of the chunks before this one.  A workgroup only waits for workgroups with a LOWER
  // index, which are dispatched before it and publish without waiting for anybody -- the wait always ends.  (Bounded
  // all the same: a broken invariant should fail a test, not hang the device.)  The three fields are summed separately:
  // a scene may hold more than 2^15 small bins.
  unsigned long long a_pts = 0;
  unsigned a_s = 0, a_b = 0;
  for (int c = tid; c < chunk; c += 256) {
    unsigned long long x = 0;
    for (int polls = 0; polls < (1 << 22); ++polls) {
      x = __hip_atomic_load(part_s + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (x) break;
      __builtin_amdgcn_s_sleep(2);
    }
    // a chunk that never published: the prefix below would be silently wrong -- K4b turns the flag into voxel_num = -1
    if (!x) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a_pts += (x >> 30) & 0x1ffffffffull;
    a_s += (unsigned)x & 0x7fffu;
    a_b += ((unsigned)x >> 15) & 0x7fffu;
  }
  unsigned long long acc = (a_pts << 32) | 0;   // points in the high word; the lists in a second 64-bit sum
  unsigned long long acl = ((unsigned long long)a_b << 32) | a_s;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    acc += __shfl_xor(acc, d, 64);
    acl += __shfl_xor(acl, d, 64);
  }
  __shared__ unsigned long long s_l[4];
  if ((tid & 63) == 0) {
    s_w[tid >> 6] = acc;
    s_l[tid >> 6] = acl;
  }
  __syncthreads();
  const unsigned long long pre = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  const unsigned long long prl = s_l[0] + s_l[1] + s_l[2] + s_l[3];
  mark(dbg, 1, 2);   // look-back done
  unsigned run = (unsigned)so.off[scene] + (unsigned)(pre >> 32) + (unsigned)psum;
  int ps = (int)(unsigned)prl + (plist & 0x7fff);
  int pb = (int)(unsigned)(prl >> 32) + (plist >> 15);
  uint4* small_s = small_list + (size_t)scene * bg.s_stride;
  uint4* big_s = big_list + (size_t)scene * bg.s_stride * 4;   // (up to 4 items per bin)
  unsigned start[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int sc = sc0 + b;
    start[b] = run;
    if (sc < bg.s_scene) {
      // one 16-byte record per bin {supercell, first place, points}: K4 / K5 read everything they need in ONE load
      if (tot[b] > (unsigned)kSmall) {
        // .w = parts << 8 | part: K5 workgroup `part` of `parts` writes the voxels whose cell has that x (/ y) parity
        const unsigned parts = tot[b] > (unsigned)kSplit4 ? 4u : tot[b] > (unsigned)kSplit2 ? 2u : 1u;
        for (unsigned q = 0; q < parts; ++q) big_s[pb++] = make_uint4((unsigned)sc, run, tot[b], (parts << 8) | q);
      } else if (tot[b] > 0u)
        small_s[ps++] = make_uint4((unsigned)sc, run, tot[b], 0u);
    }
    run += tot[b];
  }
  if (sc0 < bg.s_stride) {   // the place of every (supercell, xcd) group: a bin's eight groups are adjacent
#pragma unroll
    for (int x = 0; x < kXcd; ++x) {
      *reinterpret_cast<uint4*>(basex + xbase + (size_t)x * bg.s_stride + sc0) = make_uint4(start[0], start[1], start[2], start[3]);
      start[0] += v[x].x;
      start[1] += v[x].y;
      start[2] += v[x].z;
      start[3] += v[x].w;
    }
  }
  if (chunk == nchunks - 1 && tid == 255) {  // (the last thread's running list positions are the scene's totals)
    nlist[scene * 2] = (unsigned)ps;
    nlist[scene * 2 + 1] = (unsigned)pb;
  }
  mark(dbg, 1, 3);
}

// K3 ------------------------------------------------------------------------------------------------------------
template <bool STAGE>
__global__ void __launch_bounds__(256)
vox_bin_scatter_kernel(const float* __restrict__ pts, SceneOffsets so, int f, VoxGeom g, BinGeom bg,
                       const unsigned* __restrict__ pos, const unsigned* __restrict__ basex, uint2* __restrict__ meta,
                       float* __restrict__ rows, int rs, unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) float stage[];   // kTile * f + 8 floats (see K1)
  const int scene = blockIdx.y, tid = threadIdx.x;
  mark(dbg, 2, 0);
  const long long beg = so.off[scene], end = so.off[scene + 1];
  const long long row0 = beg + (long long)blockIdx.x * kTile;
  const int nrows = (int)max(0ll, min((long long)kTile, end - row0));
  if (nrows == 0) return;
  int shift = 0;
  if (STAGE) {
    shift = stage_rows(pts, row0, nrows, f, so.off[kMaxBatch], stage);
    __syncthreads();
  }
  const unsigned* basex_s = basex + (size_t)scene * bg.s_stride * kXcd;
  unsigned pv[4], lcs[4];
  size_t dst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {   // all four look-ups in flight before the first store
    const int p = tid + 256 * j;
    pv[j] = p < nrows ? pos[row0 + p] : kNone;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dst[j] = 0;
    lcs[j] = 0;
    if (pv[j] != kNone) {
      const int p = tid + 256 * j;
      const float* r = STAGE ? stage + shift + p * f : pts + (row0 + p) * f;
      int cx, cy, cz;
      cell_of(r[0], r[1], r[2], g, cx, cy, cz);  // same arithmetic as K1: same bin
      unsigned sc;
      bin_of(cx, cy, cz, bg, sc, lcs[j]);
      dst[j] = (size_t)basex_s[(size_t)(pv[j] >> 28) * bg.s_stride + sc] + (pv[j] & 0x0fffffffu);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (pv[j] == kNone) continue;
    const int p = tid + 256 * j;
    const float* r = STAGE ? stage + shift + p * f : pts + (row0 + p) * f;
    meta[dst[j]] = make_uint2((unsigned)(row0 + p), lcs[j]);
    float* o = rows + dst[j] * rs;
    for (int k = 0; k < rs; k += 4) {
      float4 v;
      v.x = r[k];  // k < f always (rs = f rounded up to 4)
      v.y = k + 1 < f ? r[k + 1] : 0.0f;
      v.z = k + 2 < f ? r[k + 2] : 0.0f;
      v.w = k + 3 < f ? r[k + 3] : 0.0f;
      *reinterpret_cast<float4*>(o + k) = v;
    }
  }
  mark(dbg, 2, 1);
}

// claim / find the slot of `key` in an LDS open-addressing table of t slots (power of two)
__device__ __forceinline__ unsigned lds_slot(unsigned* keys, unsigned t, unsigned key) {
  unsigned h = (key * 2654435761u) >> 16 & (t - 1);
  while (true) {
    const unsigned k = atomicCAS(&keys[h], kNone, key);
    if (k == kNone || k == key) return h;
    h = (h + 1) & (t - 1);
  }
}

__device__ __forceinline__ unsigned small_table_size(unsigned p) {
  unsigned t = 64;
  while (t < 2 * p) t <<= 1;
  return t;  // <= kSmallT for p <= kSmall
}

// K4 ------------------------------------------------------------------------------------------------------------
// flags[point] = 1 for the first point of every occupied cell.  Plain byte stores: the bits of a bin's first points are
// scattered over the whole per-scene bitmap (the cloud is shuffled), and one device-scope atomicOr per voxel -- 130k of
// them on ~350 cache lines -- kept some workgroups waiting 20 us for their atomics (`scripts/vox_timeline.py`).
__global__ void __launch_bounds__(256)
vox_first_kernel(SceneOffsets so, SceneWords sw, BinGeom bg, const uint4* __restrict__ small_list,
                 const uint4* __restrict__ big_list, const unsigned* __restrict__ nlist,
                 const uint2* __restrict__ meta, unsigned char* __restrict__ flags, unsigned long long* dbg) {
  __shared__ unsigned tab[kCells > 8 * kSmallT ? kCells : 8 * kSmallT];
  mark(dbg, 3, 0);
  const int scene = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned beg = (unsigned)so.off[scene];
  const unsigned nsmall = nlist[scene * 2], nbig = nlist[scene * 2 + 1];
  const size_t sbase = (size_t)scene * bg.s_stride;
  unsigned char* flags_s = flags + (size_t)sw.wb[scene] * 32;
  // items: [0, nbig) the big bins, then groups of four small bins (one per wave)
  const unsigned items = nbig + (nsmall + 3) / 4;
  for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
    if (item < nbig) {
      const uint4 rec = big_list[sbase * 4 + item];
      if (rec.w & 0xffu) continue;   // (parts 1.. of a bin that K5 splits: part 0 marks the whole bin)
      const unsigned p = rec.z, b = rec.y;
      for (int c = tid; c < bg.cells; c += 256) tab[c] = kNone;
      __syncthreads();
      // one pass over the bin's points (8 loads in flight per thread and trip): the cell table ends up holding every
      // occupied cell's FIRST point
      for (unsigned e0 = tid; e0 < p; e0 += 2048) {
        uint2 m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = e0 + 256 * j < p ? meta[b + e0 + 256 * j] : make_uint2(kNone, 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (m[j].x != kNone) atomicMin(&tab[m[j].y], m[j].x);
      }
      __syncthreads();
      for (int c = tid; c < bg.cells; c += 256) {
        const unsigned first = tab[c];
        if (first != kNone) flags_s[first - beg] = 1;
      }
      __syncthreads();
    } else {
      const unsigned bin = (item - nbig) * 4 + wave;
      unsigned* key = tab + wave * (2 * kSmallT);
      unsigned* val = key + kSmallT;
      unsigned p = 0, b = 0;
      if (bin < nsmall) {
        const uint4 rec = small_list[sbase + bin];
        p = rec.z;
        b = rec.y;
      }
      const unsigned t = small_table_size(p);
      for (unsigned s = lane; s < t; s += 64) {
        key[s] = kNone;
        val[s] = kNone;
      }
      __syncthreads();
      uint2 m[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = lane + 64 * j < p ? meta[b + lane + 64 * j] : make_uint2(kNone, 0u);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (m[j].x != kNone) atomicMin(&val[lds_slot(key, t, m[j].y)], m[j].x);
      __syncthreads();
      for (unsigned sidx = lane; sidx < t; sidx += 64) {
        const unsigned first = val[sidx];
        if (first != kNone) flags_s[first - beg] = 1;
      }
      __syncthreads();
    }
  }
  mark(dbg, 3, 1);   // bin work done
}

// K4b -----------------------------------------------------------------------------------------------------------
// flags -> the rank structure K5 reads: bits[w] = the 32 flags of word w, prefix[w] = first points of the scene before
// word w (= the voxel ids of the serial loop); rank == max_voxels marks i_break (the reference's `break`); voxel count.
// 8192 points per workgroup, chunk totals chained by decoupled look-back (see K2).
// part2[chunk] = 1 << 31 | first points of the chunk; 0 = not published yet.
__global__ void __launch_bounds__(256)
vox_rank_kernel(SceneOffsets so, SceneWords sw, const unsigned char* __restrict__ flags, unsigned* __restrict__ part2,
                int nchunks2, unsigned* __restrict__ bits, unsigned* __restrict__ prefix, int* __restrict__ voxel_num,
                unsigned* __restrict__ i_break, int max_voxels, unsigned* __restrict__ err, unsigned long long* dbg) {
  __shared__ int smem[17];
  __shared__ unsigned s_w[4];
  mark(dbg, 5, 0);
  const int scene = blockIdx.y, tid = threadIdx.x, chunk = blockIdx.x;
  const unsigned beg = (unsigned)so.off[scene];
  const int nw = sw.wb[scene + 1] - sw.wb[scene];
  const int w = chunk * 256 + tid;                           // this thread's word of the scene
  unsigned word = 0u;
  if (w < nw) {
    const uint4* f = reinterpret_cast<const uint4*>(flags + ((size_t)sw.wb[scene] + w) * 32);
    const uint4 f0 = f[0], f1 = f[1];
    const unsigned q[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i)   // flag bytes are 0 / 1: gather bit 0 of the four bytes of every dword
      word |= (((q[i] & 1u) | ((q[i] >> 7) & 2u) | ((q[i] >> 14) & 4u) | ((q[i] >> 21) & 8u))) << (4 * i);
  }
  const int c = __popc(word);
  int tot;
  const int pre_in = block_exclusive_scan(c, smem, &tot);
  unsigned* part_s = part2 + (size_t)scene * nchunks2;
  if (tid == 0) __hip_atomic_store(part_s + chunk, 0x80000000u | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned acc = 0;
  for (int cc = tid; cc < chunk; cc += 256) {   // look-back: lower-indexed workgroups publish without waiting (see K2)
    unsigned x = 0;
    for (int polls = 0; polls < (1 << 22); ++polls) {
      x = __hip_atomic_load(part_s + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (x) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!x) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (see K2)
    acc += x & 0x7fffffffu;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((tid & 63) == 0) s_w[tid >> 6] = acc;
  __syncthreads();
  const int run = (int)(s_w[0] + s_w[1] + s_w[2] + s_w[3]) + pre_in;
  if (w < nw) {
    bits[sw.wb[scene] + w] = word;
    prefix[sw.wb[scene] + w] = (unsigned)run;
    if (run <= max_voxels && max_voxels < run + c) {  // the first point of voxel number max_voxels: the `break`
      unsigned y = word;
      for (int i = run; i < max_voxels; ++i) y &= y - 1;
      i_break[scene] = beg + (unsigned)w * 32u + (unsigned)__ffs(y) - 1u;
    }
  }
  if (chunk == nchunks2 - 1 && tid == 255) {   // (run + c of the scene's last word = its number of first points)
    const int total = run + c;
    // The scene's last chunk has looked back over EVERY chunk of K4b, and K2 finished before this launch began: a look-back
    // of the call that ran out of polls (a broken dispatch-order invariant) is visible here.  The call then reports -1
    // voxels for the scene -- the write kernel skips it, the host raises -- instead of wrong voxel ids.
    const bool bad = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    voxel_num[scene] = bad ? -1 : min(total, max_voxels);
    if (total <= max_voxels) i_break[scene] = kNone;
  }
  mark(dbg, 5, 1);
}

// One voxel: the cell's points are seg_idx[start .. start + n) (point indices) / seg_e (their places in the bin).
struct WriteArgs {
  const float* rows;
  const unsigned* bits_s;
  const unsigned* prefix_s;
  float* voxels;
  int* coors;
  int* npv;
  float* mean;
  int f, rs, max_points, coors_cols, out_base, scene;
  unsigned beg;
};

__device__ __forceinline__ void voxel_coords(const WriteArgs& a, const BinGeom& bg, unsigned sc, unsigned lc, long long vid,
                                             int kept) {
  const int bx = sc % bg.nsx, by = (sc / bg.nsx) % bg.nsy, bz = sc / (bg.nsx * bg.nsy);
  int* c = a.coors + vid * a.coors_cols;
  if (a.coors_cols == 4) *c++ = a.scene;
  c[0] = bz * bg.sz + (int)(lc >> (2 * kSB));
  c[1] = by * kSXY + (int)((lc >> kSB) & (kSXY - 1));
  c[2] = bx * kSXY + (int)(lc & (kSXY - 1));
  a.npv[vid] = kept;
}

// 16-byte stores at 4-byte alignment (a voxel's block starts at vid * max_points * f floats): gfx950 global memory
// takes unaligned dwordx4 accesses; the packed type makes the compiler emit them.
struct __attribute__((packed, aligned(4))) f4u {
  float x, y, z, w;
};
struct __attribute__((packed, aligned(4))) i4u {
  int x, y, z, w;
};

// The common shapes (max_points == KMAX, f == F, both compile-time: ConQueR / Voxel-DETR 5 x 5, CenterPoint 4-sweep
// 5 x 6) written with 16-byte stores: 7 + 2 + 1 store instructions per voxel instead of 25 + 5 + 4 four-byte ones.
// A lane's scattered store costs the address pipeline one cache line whatever its width; with the four-byte stores the
// output stores were a third of the kernel (a build without them: 39 -> 27 us).
template <int KMAX, int F>
__device__ __forceinline__ void emit_voxel_static(const WriteArgs& a, const BinGeom& bg, unsigned sc, unsigned lc,
                                                  unsigned bin_base, const unsigned* seg_idx, const unsigned* seg_e,
                                                  unsigned start, unsigned n) {
  static_assert(F >= 4 && F <= 8, "row = one or two 16-byte pieces");
  const int kept = (int)min(n, (unsigned)KMAX);
  unsigned bi[KMAX], be[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    bi[s] = kNone;
    be[s] = 0u;
  }
#pragma unroll 4
  for (unsigned j = start; j < start + n; ++j) {
    unsigned v = seg_idx[j], e = seg_e[j];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
      const bool lt = v < bi[s];
      const unsigned tv = lt ? bi[s] : v, te = lt ? be[s] : e;
      bi[s] = lt ? v : bi[s];
      be[s] = lt ? e : be[s];
      v = tv;
      e = te;
    }
  }
  const unsigned jf = bi[0] - a.beg;
  const unsigned pw = a.prefix_s[jf >> 5], bw = a.bits_s[jf >> 5];
  float4 r0[KMAX], r1[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const float4* r = reinterpret_cast<const float4*>(a.rows + ((size_t)bin_base + be[k < kept ? k : 0]) * 8);   // rs == 8
    r0[k] = r[0];
    r1[k] = F > 4 ? r[1] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  const long long vid = a.out_base + (long long)(pw + __popc(bw & ((1u << (jf & 31)) - 1u)));
  // coordinates / count
  {
    const int bx = sc % bg.nsx, by = (sc / bg.nsx) % bg.nsy, bz = sc / (bg.nsx * bg.nsy);
    const int cz = bz * bg.sz + (int)(lc >> (2 * kSB)), cy = by * kSXY + (int)((lc >> kSB) & (kSXY - 1)),
              cx = bx * kSXY + (int)(lc & (kSXY - 1));
    if (a.coors_cols == 4) {
      i4u c4;
      c4.x = a.scene, c4.y = cz, c4.z = cy, c4.w = cx;
      *reinterpret_cast<i4u*>(a.coors + vid * 4) = c4;
    } else {
      int* c = a.coors + vid * 3;
      c[0] = cz, c[1] = cy, c[2] = cx;
    }
    a.npv[vid] = kept;
  }
  // the voxel's KMAX x F block, flat, rows past `kept` zero
  float flat[KMAX * F];
  float acc[F];
#pragma unroll
  for (int t = 0; t < F; ++t) acc[t] = 0.0f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const float rv[8] = {r0[k].x, r0[k].y, r0[k].z, r0[k].w, r1[k].x, r1[k].y, r1[k].z, r1[k].w};
#pragma unroll
    for (int t = 0; t < F; ++t) {
      const float v = k < kept ? rv[t] : 0.0f;
      flat[k * F + t] = v;
      acc[t] = __fadd_rn(acc[t], v);   // slot order, like the reader's sum(dim=1); + 0.0f past `kept` changes nothing
    }
  }
  {
    float* o = a.voxels + vid * (KMAX * F);
#pragma unroll
    for (int q = 0; q + 4 <= KMAX * F; q += 4) {
      f4u v4;
      v4.x = flat[q], v4.y = flat[q + 1], v4.z = flat[q + 2], v4.w = flat[q + 3];
      *reinterpret_cast<f4u*>(o + q) = v4;
    }
#pragma unroll
    for (int q = (KMAX * F) & ~3; q < KMAX * F; ++q) o[q] = flat[q];
  }
  if (a.mean) {
    float* mo = a.mean + vid * F;
    const float d = (float)kept;
    f4u m4;
    m4.x = __fdiv_rn(acc[0], d), m4.y = __fdiv_rn(acc[1], d), m4.z = __fdiv_rn(acc[2], d), m4.w = __fdiv_rn(acc[3], d);
    *reinterpret_cast<f4u*>(mo) = m4;
#pragma unroll
    for (int t = 4; t < F; ++t) mo[t] = __fdiv_rn(acc[t], d);
  }
}

// FAST (KMAX = 5 or 8 >= max_points, f <= 8: every configuration of the reference's playground): the max_points lowest point
// indices of the cell by streaming insertion into a sorted register array -- one pass over the segment -- then ALL
// global loads of the voxel (bitmap word, prefix word, up to eight 32-byte rows; rows past `kept` re-read row 0) issued
// together and branch-free, so the voxel costs ONE memory round trip before its stores (loads inside `if (k < kept)`
// were waited for one row at a time: five round trips per voxel).  The mean comes from registers.
// Generic: selection by repeated minimum (O(n * kept) dependent loads), rows and mean through memory.
template <int KMAX>   // 0: generic path; 5 / 8: FAST with max_points <= KMAX
__device__ __forceinline__ void emit_voxel(const WriteArgs& a, const BinGeom& bg, unsigned sc, unsigned lc,
                                           unsigned bin_base, const unsigned* seg_idx, const unsigned* seg_e,
                                           unsigned start, unsigned n) {
  const int kept = (int)min(n, (unsigned)a.max_points);
  if constexpr (KMAX > 0) {
    unsigned bi[KMAX], be[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
      bi[s] = kNone;
      be[s] = 0u;
    }
#pragma unroll 4
    for (unsigned j = start; j < start + n; ++j) {
      unsigned v = seg_idx[j], e = seg_e[j];
#pragma unroll
      for (int s = 0; s < KMAX; ++s) {
        const bool lt = v < bi[s];
        const unsigned tv = lt ? bi[s] : v, te = lt ? be[s] : e;
        bi[s] = lt ? v : bi[s];
        be[s] = lt ? e : be[s];
        v = tv;
        e = te;
      }
    }
    const unsigned jf = bi[0] - a.beg;
    const unsigned pw = a.prefix_s[jf >> 5], bw = a.bits_s[jf >> 5];
    float4 r0[KMAX], r1[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const float4* r = reinterpret_cast<const float4*>(a.rows + ((size_t)bin_base + be[k < kept ? k : 0]) * a.rs);
      r0[k] = r[0];
      r1[k] = a.rs > 4 ? r[1] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // voxel id = number of first points of the scene before this voxel's first point
    const long long vid = a.out_base + (long long)(pw + __popc(bw & ((1u << (jf & 31)) - 1u)));
    voxel_coords(a, bg, sc, lc, vid, kept);
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.0f;
    float* o = a.voxels + vid * a.max_points * a.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < kept) {
#define EFG_VOX_PUT(t, val)                  \
  if (t < a.f) {                             \
    o[k * a.f + t] = val;                    \
    acc[t] = __fadd_rn(acc[t], val); /* slot order, like the reader's sum(dim=1) */ \
  }
        EFG_VOX_PUT(0, r0[k].x)
        EFG_VOX_PUT(1, r0[k].y)
        EFG_VOX_PUT(2, r0[k].z)
        EFG_VOX_PUT(3, r0[k].w)
        EFG_VOX_PUT(4, r1[k].x)
        EFG_VOX_PUT(5, r1[k].y)
        EFG_VOX_PUT(6, r1[k].z)
        EFG_VOX_PUT(7, r1[k].w)
#undef EFG_VOX_PUT
      }
    }
    for (int k = kept * a.f; k < a.max_points * a.f; ++k) o[k] = 0.0f;
    if (a.mean) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < a.f) a.mean[vid * a.f + t] = __fdiv_rn(acc[t], (float)kept);
    }
  } else {
    long long vid = 0;
    unsigned prev = 0;
    for (int k = 0; k < kept; ++k) {
      unsigned best = kNone, be = 0;
      for (unsigned j = start; j < start + n; ++j) {
        const unsigned v = seg_idx[j];
        if ((k == 0 || v > prev) && v < best) {
          best = v;
          be = seg_e[j];
        }
      }
      prev = best;
      if (k == 0) {
        const unsigned jf = best - a.beg;
        vid = a.out_base + (long long)(a.prefix_s[jf >> 5] + __popc(a.bits_s[jf >> 5] & ((1u << (jf & 31)) - 1u)));
        voxel_coords(a, bg, sc, lc, vid, kept);
      }
      const float* r = a.rows + ((size_t)bin_base + be) * a.rs;
      float* o = a.voxels + (vid * a.max_points + k) * a.f;
      for (int t = 0; t < a.f; ++t) o[t] = r[t];
    }
    float* o = a.voxels + vid * a.max_points * a.f;
    for (int k = kept * a.f; k < a.max_points * a.f; ++k) o[k] = 0.0f;
    if (a.mean) {
      for (int t = 0; t < a.f; ++t) {
        float sm = 0.0f;
        for (int k = 0; k < kept; ++k) sm = __fadd_rn(sm, o[k * a.f + t]);
        a.mean[vid * a.f + t] = __fdiv_rn(sm, (float)kept);
      }
    }
  }
}

// K5 ------------------------------------------------------------------------------------------------------------
// LDS: a big bin uses [cell table | point list | voxel list], four small bins use 4 x [keys | offsets | point list | voxel list]
constexpr int kSmallLds = 2 * kSmallT + 2 * kSmall + kSmallT;   // words per wave
constexpr int kBigLds = 2 * kCells + 2 * kSegLds;
constexpr int kWriteLds = kBigLds > 4 * kSmallLds ? kBigLds : 4 * kSmallLds;

template <int KMAX, int F>   // F > 0: max_points == KMAX and f == F exactly (16-byte stores); F == 0: run-time sizes
__global__ void __launch_bounds__(256)
vox_write_kernel(SceneOffsets so, SceneWords sw, BinGeom bg, int f, int rs, const uint4* __restrict__ small_list,
                 const uint4* __restrict__ big_list, const unsigned* __restrict__ nlist,
                 const uint2* __restrict__ meta, const float* __restrict__ rows, const unsigned* __restrict__ bits,
                 const unsigned* __restrict__ prefix, const unsigned* __restrict__ i_break,
                 const int* __restrict__ voxel_num, int max_points, int coors_cols, float* __restrict__ voxels,
                 int* __restrict__ coors, int* __restrict__ npv, float* __restrict__ mean,
                 unsigned* __restrict__ seg_idx_g, unsigned* __restrict__ seg_e_g, size_t seg_stride,
                 unsigned long long* dbg) {
  __shared__ unsigned tab[kWriteLds];
  __shared__ int smem[17];
  mark(dbg, 4, 0);
  const int scene = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned nsmall = nlist[scene * 2], nbig = nlist[scene * 2 + 1];
  const size_t sbase = (size_t)scene * bg.s_stride;
  const unsigned items = nbig + (nsmall + 3) / 4;
  if (blockIdx.x >= items) return;
  WriteArgs a;
  a.rows = rows;
  a.bits_s = bits + sw.wb[scene];
  a.prefix_s = prefix + sw.wb[scene];
  a.voxels = voxels;
  a.coors = coors;
  a.npv = npv;
  a.mean = mean;
  a.f = f;
  a.rs = rs;
  a.max_points = max_points;
  a.coors_cols = coors_cols;
  a.scene = scene;
  a.beg = (unsigned)so.off[scene];
  a.out_base = 0;
  for (int b = 0; b <= scene; ++b) {
    if (voxel_num[b] < 0) return;   // a look-back of this call failed (K2 / K4b): nothing is written, the host raises
    if (b < scene) a.out_base += voxel_num[b];
  }
  const unsigned ib = i_break[scene];  // points from here on are never processed (voxelization_cpu.cpp:78-79)
  // Items are dealt in a zig-zag (w, 2G-1-w, 2G+w, ...): the big bins come first in the list, so the workgroups that
  // hold them (the slowest of a round) take their next item LAST and from the cheap end.
  for (unsigned rnd = 0, item = blockIdx.x; item < items;
       ++rnd, item = (rnd & 1) ? (rnd + 1) * gridDim.x - 1 - blockIdx.x : rnd * gridDim.x + blockIdx.x) {
    if (item < nbig) {
      unsigned* seg_lds = tab + kCells;
      unsigned* vlist = tab + kCells + 2 * kSegLds;   // occupied cells, compacted
      const uint4 rec = big_list[sbase * 4 + item];
      const unsigned sc = rec.x, p = rec.z, b = rec.y;
      // this workgroup's share of the bin's cells: all (parts 1), x parity (2), x and y parity (4); lc = (z, y, x) with
      // kSB bits each for y and x
      const unsigned parts = rec.w >> 8, part = rec.w & 0xffu;
      const unsigned pmask = parts == 4 ? (1u | (1u << kSB)) : parts == 2 ? 1u : 0u;
      const unsigned pval = parts == 4 ? ((part & 1u) | ((part >> 1) << kSB)) : part;
      for (int c = tid; c < bg.cells; c += 256) tab[c] = 0u;
      __syncthreads();
      const bool in_reg = p <= 4 * 256;   // the bin's points stay in registers between the count and the scatter pass
      uint2 mr[4];
      if (in_reg) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mr[j] = tid + 256 * j < p ? meta[b + tid + 256 * j] : make_uint2(kNone, 0u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (mr[j].x < ib && (mr[j].y & pmask) == pval) atomicAdd(&tab[mr[j].y], 1u);   // (kNone >= ib always)
      } else {
        for (unsigned e0 = tid; e0 < p; e0 += 1024) {
          uint2 m[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = e0 + 256 * j < p ? meta[b + e0 + 256 * j] : make_uint2(kNone, 0u);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (m[j].x < ib && (m[j].y & pmask) == pval) atomicAdd(&tab[m[j].y], 1u);
        }
      }
      __syncthreads();
      int nvox, tot;
      {  // exclusive scan over the cells in place (thread t owns cells [t * per, (t + 1) * per)) and, in the same pass,
         // the list of occupied cells in cell order
        const int per = (bg.cells + 255) / 256, c0 = min(tid * per, bg.cells), c1 = min(c0 + per, bg.cells);
        int sum = 0, occ = 0;
        for (int c = c0; c < c1; ++c) {
          sum += (int)tab[c];
          occ += tab[c] ? 1 : 0;
        }
        int run = block_exclusive_scan(sum, smem, &tot);
        int vpos = block_exclusive_scan(occ, smem, &nvox);
        for (int c = c0; c < c1; ++c) {
          const int v = (int)tab[c];
          tab[c] = (unsigned)run;
          run += v;
          if (v) vlist[vpos++] = (unsigned)c;
        }
      }
      __syncthreads();
      // the bin's points sorted by cell: in LDS when they fit, else in the bin's own range of a global scratch
      const bool in_lds = tot <= kSegLds;   // (this workgroup's share of the bin)
      unsigned* si = in_lds ? seg_lds : seg_idx_g + (size_t)part * seg_stride + b;
      unsigned* se = in_lds ? seg_lds + kSegLds : seg_e_g + (size_t)part * seg_stride + b;
      if (in_reg) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (mr[j].x < ib && (mr[j].y & pmask) == pval) {
            const unsigned q = atomicAdd(&tab[mr[j].y], 1u);  // afterwards tab[c] = END of cell c = start of cell c + 1
            si[q] = mr[j].x;
            se[q] = tid + 256 * j;
          }
        }
      } else {
        for (unsigned e0 = tid; e0 < p; e0 += 1024) {
          uint2 m[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = e0 + 256 * j < p ? meta[b + e0 + 256 * j] : make_uint2(kNone, 0u);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (m[j].x < ib && (m[j].y & pmask) == pval) {
              const unsigned q = atomicAdd(&tab[m[j].y], 1u);
              si[q] = m[j].x;
              se[q] = e0 + 256 * j;
            }
          }
        }
      }
      __syncthreads();
      mark(dbg, 4, 1);   // sorted by cell
      for (int v = tid; v < nvox; v += 256) {
        const unsigned c = vlist[v];
        const unsigned en = tab[c], st = c ? tab[c - 1] : 0u;
        if constexpr (F > 0) emit_voxel_static<KMAX, F>(a, bg, sc, c, b, si, se, st, en - st);
        else emit_voxel<KMAX>(a, bg, sc, c, b, si, se, st, en - st);
      }
      __syncthreads();
      mark(dbg, 4, 2);   // big bin written
    } else {
      const unsigned bin = (item - nbig) * 4 + wave;
      unsigned* key = tab + wave * kSmallLds;
      unsigned* off = key + kSmallT;
      unsigned* si = off + kSmallT;
      unsigned* se = si + kSmall;
      unsigned* vlist = se + kSmall;
      unsigned p = 0, b = 0, sc = 0;
      if (bin < nsmall) {
        const uint4 rec = small_list[sbase + bin];
        sc = rec.x;
        p = rec.z;
        b = rec.y;
      }
      const unsigned t = small_table_size(p);
      for (unsigned s = lane; s < t; s += 64) {
        key[s] = kNone;
        off[s] = 0u;
      }
      __syncthreads();
      uint2 m[4];
      unsigned sl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = lane + 64 * j < p ? meta[b + lane + 64 * j] : make_uint2(kNone, 0u);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sl[j] = 0;
        if (m[j].x < ib) {   // (kNone >= ib always)
          sl[j] = lds_slot(key, t, m[j].y);
          atomicAdd(&off[sl[j]], 1u);
        }
      }
      __syncthreads();
      int nvox;
      {  // exclusive scan over the t slots of this wave's table (lane l owns slots [l * t / 64, (l + 1) * t / 64)) and the
         // list of occupied slots: positions from the wave's inclusive scan of the per-lane counts
        const unsigned per = t >> 6, s0 = lane * per;
        int sum = 0, occ = 0;
        for (unsigned s = 0; s < per; ++s) {
          sum += (int)off[s0 + s];
          occ += off[s0 + s] ? 1 : 0;
        }
        int run = wave_inclusive_scan(sum) - sum;
        const int vinc = wave_inclusive_scan(occ);
        int vpos = vinc - occ;
        nvox = __shfl(vinc, 63, 64);
        for (unsigned s = 0; s < per; ++s) {
          const int v = (int)off[s0 + s];
          off[s0 + s] = (unsigned)run;
          run += v;
          if (v) vlist[vpos++] = s0 + s;
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (m[j].x < ib) {
          const unsigned q = atomicAdd(&off[sl[j]], 1u);
          si[q] = m[j].x;
          se[q] = lane + 64 * j;
        }
      }
      __syncthreads();
      mark(dbg, 4, 3);   // small bins sorted
      for (int v = lane; v < nvox; v += 64) {
        const unsigned s = vlist[v];
        const unsigned en = off[s], st = s ? off[s - 1] : 0u;
        if constexpr (F > 0) emit_voxel_static<KMAX, F>(a, bg, sc, key[s], b, si, se, st, en - st);
        else emit_voxel<KMAX>(a, bg, sc, key[s], b, si, se, st, en - st);
      }
      __syncthreads();
      mark(dbg, 4, 4);   // small bins written
    }
  }
}

struct BinsLayout {
  BinGeom bg;
  size_t s_total;
  int rs, nchunks;
  int64_t n;
};

bool bins_layout(int64_t n_total, int batch, int f, const VoxGeom& g, BinsLayout* L) {
  L->bg = bin_geom(g);
  if (L->bg.s_scene < 0) return false;
  L->s_total = (size_t)L->bg.s_stride * batch;
  // a grid with far more supercells than points (or than 4M) is not worth binning, and K1 packs a point's place inside
  // its group into 28 bits: the hash path takes the rest
  if (L->s_total > (size_t)std::max<int64_t>(1 << 18, 8 * n_total) || L->s_total > (1u << 22)) return false;
  if (n_total >= (1ll << 28)) return false;
  L->rs = (f + 3) / 4 * 4;
  L->nchunks = (int)ceil_div((int64_t)L->bg.s_stride, kScanChunk);
  L->n = std::max<int64_t>(n_total, 1);
  return true;
}

// The words of a call that must be ZERO when it starts -- the per-XCD counters, the look-back records of the two scans, the
// error word -- live in a buffer the LIBRARY owns per (device, stream): K2, the only reader of the counters, leaves them zero for
// the next call, and K1 zeroes the few hundred words of records + error word that only later launches of its own call use -- so a
// call needs no clear launch (7 -> 6 launches, ~4.5 us of a 2 x 180k call).  Allocated with hipMalloc + hipMemset on first use and when a call needs
// more words than it holds (both synchronise: once per process and size); a call that fails between its launches leaves the
// buffer marked dirty and the next call zeroes it with one asynchronous memset.  Calls on ONE
// stream are issued by one thread at a time (as for stream-K, include/efg_hip.h).  While the stream is being CAPTURED into a HIP
// graph the call takes its cleared words from the caller's workspace and clears them with vox_clear_kernel, as before (a
// replay must not depend on what eager calls left behind, and the first-use allocation would invalidate the capture).
constexpr size_t kOwnRecordWords = 49152;   // >= the records of any call the binned path takes (<= 2^22 supercells, < 2^28 points)
struct VoxState {
  unsigned* words = nullptr;
  size_t cap = 0;
  bool dirty = true;
};
std::mutex g_vox_mu;
std::map<std::pair<int, hipStream_t>, VoxState> g_vox_pool;

constexpr size_t kVoxPoolMax = 32;   // (device, stream) entries kept; a process that keeps creating streams recycles them

int vox_state_for_stream(hipStream_t stream, size_t words, VoxState** out) {
  int dev = 0;
  EFG_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_vox_mu);
  const auto key = std::make_pair(dev, stream);
  // hipMalloc / hipFree below synchronise the device; under a capture in progress on ANOTHER stream (global capture mode) that
  // would invalidate it: this thread's allocation calls run in relaxed mode, which the other thread's capture tolerates
  struct RelaxedCapture {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    bool on = false;
    void enter() {
      if (!on) on = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    }
    ~RelaxedCapture() {
      if (on) (void)hipThreadExchangeStreamCaptureMode(&mode);
    }
  } relaxed;
  if (!g_vox_pool.count(key) && g_vox_pool.size() >= kVoxPoolMax) {
    // streams that were destroyed (or are simply many): drop every other entry (hipFree synchronises, so no launch still
    // reads a buffer that goes); a stream handle that is reused later starts from a fresh, zeroed buffer
    relaxed.enter();
    for (auto it = g_vox_pool.begin(); it != g_vox_pool.end();) {
      if (it->second.words) (void)hipFree(it->second.words);
      it = g_vox_pool.erase(it);
    }
  }
  VoxState& st = g_vox_pool[key];
  if (st.cap < words) {
    relaxed.enter();
    if (st.words) EFG_HIP_TRY(hipFree(st.words));   // (synchronises: nothing of this stream still reads it)
    st.words = nullptr;
    st.cap = 0;
    const size_t cap = align_up(words + words / 4, 1024);
    EFG_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st.words), cap * sizeof(unsigned)));
    st.cap = cap;
    st.dirty = true;
  }
  if (st.dirty) EFG_HIP_TRY(hipMemsetAsync(st.words, 0, st.cap * sizeof(unsigned), stream));
  st.dirty = true;   // until the call's last launch has been issued (vox_state_done)
  *out = &st;
  return EFG_OK;
}

void vox_state_done(VoxState* st) {   // every launch of the call is queued: the counters are zero again when K2 has run
  std::lock_guard<std::mutex> lock(g_vox_mu);
  st->dirty = false;
}

}  // namespace

void bins_set_debug_timeline(unsigned long long* buf) { g_dbg = buf; }
size_t bins_debug_timeline_words() { return (size_t)6 * 8 * kDbgMaxWg; }

size_t bins_workspace_bytes(int64_t n_total, int batch, int f, const VoxGeom& g) {
  BinsLayout L;
  if (!bins_layout(n_total, batch, f, g, &L)) return 0;
  const size_t words = (size_t)(n_total / 32 + 4 * batch + 4);
  size_t b = 0;
  b += align_up((L.s_total * kXcd + (size_t)L.nchunks * batch * 2 + ((size_t)(n_total / 8192 + 2) * batch) + 4) * 4, 256);  // count, part, part2, err (cleared)
  b += align_up(L.s_total * kXcd * 4, 256);                    // basex
  b += 5 * align_up(L.s_total * 16, 256);                      // small_list, big_list x4 (16-byte bin records)
  b += align_up(4 * kMaxBatch * 4, 256);                       // nlist, i_break
  b += align_up((size_t)L.n * 4, 256);                         // pos
  b += align_up((size_t)L.n * 8, 256);                         // meta
  b += align_up((size_t)L.n * L.rs * 4, 256);                  // rows
  b += 2 * align_up((size_t)L.n * 16, 256);                    // seg_idx, seg_e (bins above kSegLds points; x4: parts of split bins)
  b += 2 * align_up(words * 4, 256);                           // bits, prefix
  b += align_up(words * 32, 256);                              // flags (one byte per point)
  return b + 256;
}

int bins_hard_voxelize(const HardArgs& a) {
  hipStream_t stream = a.stream;
  BinsLayout L;
  if (!bins_layout(a.n_total, a.batch, a.f, a.g, &L)) {
    set_error("hard_voxelize: grid cannot be binned");
    return EFG_E_INVALID;
  }
  const int batch = a.batch, f = a.f;
  EFG_CHECK_ARG((long long)batch * a.max_voxels * a.max_points < (1ll << 31), "voxel capacity too large");
  SceneWords sw;
  SceneOffsets so = a.so;
  sw.wb[0] = 0;
  for (int b = 0; b < batch; ++b)   // every scene's words start on a 16-byte boundary (vector loads of the scan)
    sw.wb[b + 1] = sw.wb[b] + (int)(ceil_div(so.off[b + 1] - so.off[b], 128) * 4);
  for (int b = batch + 1; b <= kMaxBatch; ++b) so.off[b] = so.off[batch];  // off[kMaxBatch] = total rows (stage_rows)
  const size_t words = (size_t)(a.n_total / 32 + 4 * batch + 4);
  Workspace w(a.ws, a.ws_bytes);
  int max_words = 0;
  for (int b = 0; b < batch; ++b) max_words = std::max(max_words, sw.wb[b + 1] - sw.wb[b]);
  const int nchunks2 = std::max(1, (int)ceil_div(max_words, 256));   // K4b: 256 words = 8192 points per workgroup
  const size_t part2_words = (size_t)(a.n_total / 8192 + 2) * batch;   // >= nchunks2 * batch
  const size_t cleared_words = L.s_total * kXcd + (size_t)L.nchunks * batch * 2 + part2_words + 4;   // (+ the error word)
  unsigned* count = w.take<unsigned>(cleared_words);   // (the workspace keeps its layout whether or not these words are used)
  // (8-byte aligned: s_total is a multiple of 4)
  unsigned long long* part = count ? reinterpret_cast<unsigned long long*>(count + L.s_total * kXcd) : nullptr;
  unsigned* part2 = count ? count + L.s_total * kXcd + (size_t)L.nchunks * batch * 2 : nullptr;
  unsigned* err = count ? count + cleared_words - 4 : nullptr;
  unsigned* basex = w.take<unsigned>(L.s_total * kXcd);
  uint4* small_list = w.take<uint4>(L.s_total);
  uint4* big_list = w.take<uint4>(4 * L.s_total);
  unsigned* nlist = w.take<unsigned>(4 * kMaxBatch);
  unsigned* i_break = nlist ? nlist + 2 * kMaxBatch : nullptr;
  unsigned* pos = w.take<unsigned>(L.n);
  uint2* meta = w.take<uint2>(L.n);
  float* rows = w.take<float>((size_t)L.n * L.rs);
  unsigned* seg_idx = w.take<unsigned>(4 * (size_t)L.n);   // (one range per part of a split bin)
  unsigned* seg_e = w.take<unsigned>(4 * (size_t)L.n);
  unsigned* bits = w.take<unsigned>(words);
  unsigned* prefix = w.take<unsigned>(words);
  unsigned char* flags = w.take<unsigned char>(words * 32);
  if (!w.ok) {
    set_error("hard_voxelize workspace too small: need %zu bytes, got %zu",
              bins_workspace_bytes(a.n_total, batch, f, a.g), a.ws_bytes);
    return EFG_E_WORKSPACE;
  }
  const dim3 blk(256);
  // the cleared words: the library's own, self-cleaning buffer of this stream (see VoxState) unless the stream is being captured
  VoxState* own = nullptr;
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing(stream, &cap);
    const size_t record_words = cleared_words - L.s_total * kXcd;
    if (cap == hipStreamCaptureStatusNone && record_words <= kOwnRecordWords) {
      // [records: kOwnRecordWords, zeroed by K1 of the call that uses them][counters: zero at rest].  The records sit at a
      // FIXED place in front: behind the counters their offset would change with the grid and the batch, and the stale records
      // of one call would be the next layout's counters.
      if (int rc = vox_state_for_stream(stream, kOwnRecordWords + L.s_total * kXcd, &own)) return rc;
      part = reinterpret_cast<unsigned long long*>(own->words);
      part2 = own->words + (size_t)L.nchunks * batch * 2;
      err = own->words + record_words - 4;
      count = own->words + kOwnRecordWords;
    }
  }
  if (!own) {
    // (a kernel, not hipMemsetAsync: captured into a HIP graph the memset NODE of this call did not take effect on the second
    // replay -- counters left dirty, wild offsets, "write access to a read-only page"; scripts/ubench/vox_graph_probe.py)
    hipLaunchKernelGGL(vox_clear_kernel, dim3((unsigned)std::min<size_t>(ceil_div((int64_t)cleared_words, 1024), 1024)), blk, 0, stream,
                       reinterpret_cast<unsigned*>(count), (size_t)cleared_words);
  }
  const int tiles = (int)std::max<int64_t>(1, ceil_div(a.max_scene, kTile));
  const bool stage = f <= 8;
  const size_t stage_bytes = stage ? sizeof(float) * ((size_t)kTile * f + 8) : 16;
  if (stage)
    hipLaunchKernelGGL(vox_bin_count_kernel<true>, dim3(tiles, batch), blk, stage_bytes, stream, a.points, so, sw, f, a.g, L.bg, count,
                       pos, flags, own ? reinterpret_cast<unsigned*>(part) : nullptr, (int)(cleared_words - L.s_total * kXcd), g_dbg);
  else
    hipLaunchKernelGGL(vox_bin_count_kernel<false>, dim3(tiles, batch), blk, stage_bytes, stream, a.points, so, sw, f, a.g, L.bg, count,
                       pos, flags, own ? reinterpret_cast<unsigned*>(part) : nullptr, (int)(cleared_words - L.s_total * kXcd), g_dbg);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(vox_bin_scan_kernel, dim3(L.nchunks, batch), blk, 0, stream, so, L.bg, count, part, L.nchunks, basex,
                     small_list, big_list, nlist, err, own ? 1 : 0, g_dbg);
  EFG_LAUNCH_CHECK();
  if (a.n_total > 0) {
    if (stage)
      hipLaunchKernelGGL(vox_bin_scatter_kernel<true>, dim3(tiles, batch), blk, stage_bytes, stream, a.points, so, f, a.g, L.bg, pos,
                         basex, meta, rows, L.rs, g_dbg);
    else
      hipLaunchKernelGGL(vox_bin_scatter_kernel<false>, dim3(tiles, batch), blk, stage_bytes, stream, a.points, so, f, a.g, L.bg, pos,
                         basex, meta, rows, L.rs, g_dbg);
    EFG_LAUNCH_CHECK();
  }
  // the bins are listed on the device; a fixed grid loops over them (bounded by what the host knows: a bin holds a point)
  const int64_t items_ub = std::min<int64_t>(4 * (int64_t)L.bg.s_scene, std::max<int64_t>(a.max_scene, 1));
  // as many workgroups as are resident at once (4 per CU x 256 CUs), shared by the scenes of the batch
  const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(items_ub, std::max(256, kItemGrid / batch)));
  // (K4 is light -- 16 KB of LDS, 8 workgroups per CU: every bin gets its own workgroup up to 2048 of them)
  const int gx4 = (int)std::max<int64_t>(1, std::min<int64_t>(items_ub, std::max(256, 2 * kItemGrid / batch)));
  if (a.n_total > 0) {
    hipLaunchKernelGGL(vox_first_kernel, dim3(gx4, batch), blk, 0, stream, so, sw, L.bg, small_list, big_list, nlist, meta,
                       flags, g_dbg);
    EFG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(vox_rank_kernel, dim3(nchunks2, batch), blk, 0, stream, so, sw, flags, part2, nchunks2, bits, prefix,
                     a.voxel_num, i_break, a.max_voxels, err, g_dbg);
  EFG_LAUNCH_CHECK();
  if (a.n_total > 0) {
#define EFG_VOX_WRITE(KM, FF)                                                                                            \
  hipLaunchKernelGGL((vox_write_kernel<KM, FF>), dim3(gx, batch), blk, 0, stream, so, sw, L.bg, f, L.rs, small_list,       \
                     big_list, nlist, meta, rows, bits, prefix, i_break, a.voxel_num, a.max_points, a.coors_cols, a.voxels, \
                     a.coors, a.npv, a.mean, seg_idx, seg_e, (size_t)L.n, g_dbg)
    if (a.max_points == 5 && f == 5) EFG_VOX_WRITE(5, 5);
    else if (a.max_points == 5 && f == 6) EFG_VOX_WRITE(5, 6);
    else if (a.max_points <= 5 && f <= 8) EFG_VOX_WRITE(5, 0);
    else if (a.max_points <= 8 && f <= 8) EFG_VOX_WRITE(8, 0);
    else EFG_VOX_WRITE(0, 0);
#undef EFG_VOX_WRITE
    EFG_LAUNCH_CHECK();
  }
  if (own) vox_state_done(own);
  return EFG_OK;
}

}  // namespace efg
