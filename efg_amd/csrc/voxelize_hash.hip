// Hard voxelization, generation 1 (rounds 1-3): one global open-addressing hash over all cells.
//
// Kept (a) as the A/B partner of the supercell-binned voxelizer in voxelize_bins.hip (EFG_VOX_IMPL=hash) and (b) for
// grids with so many supercells per point that binning them is pointless (bins_workspace_bytes() == 0).
//
// Parallel, bit-exact formulation of the serial first-come-first-kept loop (voxelization_cpu.cpp:43-99):
//   K1 insert   one open-addressing hash insert per point on a 64-bit {cell, point} word:
//               atomicMin keeps, per occupied cell, the LOWEST point index = first occurrence;
//   K2 count    per 1024-point tile, count the points that are the first of their voxel (+ per-scene totals);
//   K3 assign   block-scan the first-flags in POINT ORDER: voxel id = rank of its first point;
//               rank == max_voxels marks i_break (the reference's `break`); scene bases / voxel counts;
//   K4 cascade  every later point of a kept voxel inserts its index into the voxel's sorted
//               max_points-entry list with a chain of atomicMin (carry = max(old, mine));
//   K5 gather   copy the selected points, zero padding, counts and the fused per-voxel mean.
// 5 kernels + 1 memset per call; measured 148.7 us for 2 x 180k points, 8.7x the algorithmic bytes in HBM traffic
// (profiles/r03z_bench.json) -- random 64-bit device-scope atomics on a 2N-slot table, three visits per point.
#include "voxelize_common.h"

namespace efg {
namespace {

constexpr int kTile = 1024;  // points per block in the ordered stages (256 threads x 4)
constexpr unsigned long long kEmpty = ~0ull;
constexpr unsigned kInf = 0xffffffffu;  // memset(0xff) pattern: "no point" (lists / i_break are unsigned)
constexpr int kFirstFlag = 1 << 30;

__device__ __forceinline__ unsigned hash_cell(unsigned key, int shift) { return (key * 2654435761u) >> shift; }

// the hash table and the per-scene break / total words of a call: filled words
__global__ void __launch_bounds__(256) vox_fill_kernel(unsigned* __restrict__ p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n) p[i + u * 256] = v;
  }
}

// K1: slot_of_point[i] = hash slot of the point's cell (or -1); table[slot] = {cell, min point}.
__global__ void __launch_bounds__(256)
vox_insert_kernel(const float* __restrict__ pts, SceneOffsets so, int f, VoxGeom g, unsigned vol,
                  unsigned long long* __restrict__ table, unsigned tmask, int tshift, int* __restrict__ slot_of_point,
                  int precheck) {
  const int scene = blockIdx.y;
  const long long beg = so.off[scene], end = so.off[scene + 1];
  for (long long i = beg + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += (long long)gridDim.x * blockDim.x) {
    int cx, cy, cz;
    if (!point_cell(pts + i * f, g, cx, cy, cz)) {
      slot_of_point[i] = -1;
      continue;
    }
    const unsigned key = (unsigned)scene * vol + ((unsigned)cz * g.grid[1] + cy) * g.grid[0] + cx;
    const unsigned long long mine = ((unsigned long long)key << 32) | (unsigned)i;
    unsigned h = hash_cell(key, tshift);
    while (true) {
      unsigned long long cur = table[h];
      if (cur == kEmpty) {
        cur = atomicCAS(&table[h], kEmpty, mine);
        if (cur == kEmpty) break;  // claimed
      }
      if ((unsigned)(cur >> 32) == key) {
        // a slot's key never changes and its point index only decreases: an index already below mine (however stale
        // the read) means the atomic would be a no-op.  Points arrive roughly in index order, so this is the common
        // case for every later point of a voxel -- and a device-scope atomic costs a trip to the memory side.
        if (!precheck || (unsigned)cur > (unsigned)i) atomicMin(&table[h], mine);
        break;
      }
      h = (h + 1) & tmask;
    }
    slot_of_point[i] = (int)h;
  }
}

// K2: tag first points (kFirstFlag in slot_of_point) and count them per tile.
__global__ void __launch_bounds__(256)
vox_count_kernel(SceneOffsets so, const unsigned long long* __restrict__ table, int* __restrict__ slot_of_point,
                 int* __restrict__ tile_counts, int tiles_per_scene, int* __restrict__ scene_total) {
  __shared__ int smem[17];
  const int scene = blockIdx.y;
  const long long beg = so.off[scene], end = so.off[scene + 1];
  const long long base = beg + (long long)blockIdx.x * kTile + threadIdx.x * 4;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = base + j;
    if (i < end) {
      const int s = slot_of_point[i];
      if (s >= 0 && (unsigned)table[s] == (unsigned)i) {
        slot_of_point[i] = s | kFirstFlag;
        ++cnt;
      }
    }
  }
  cnt = wave_reduce_sum(cnt);
  if (lane_id() == 0) smem[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int c = smem[0] + smem[1] + smem[2] + smem[3];
    tile_counts[scene * tiles_per_scene + blockIdx.x] = c;
    if (c) atomicAdd(&scene_total[scene], c);  // starts at -1 (the common 0xff fill): readers add 1
  }
}

// K3: voxel ids in first-occurrence order; first point goes to list entry 0.
__global__ void __launch_bounds__(256)
vox_assign_kernel(SceneOffsets so, const unsigned long long* __restrict__ table, const int* __restrict__ slot_of_point,
                  const int* __restrict__ tile_counts, int tiles_per_scene, const int* __restrict__ scene_total,
                  int batch, int* __restrict__ scene_base, int* __restrict__ voxel_num,
                  int max_voxels, int max_points, unsigned vol, VoxGeom g, int* __restrict__ vid_of_slot,
                  unsigned* __restrict__ lists, unsigned* __restrict__ i_break, int* __restrict__ coors, int coors_cols) {
  __shared__ int smem[17];
  __shared__ int s_prefix;
  const int scene = blockIdx.y;
  // output base row of this scene = kept voxels of the scenes before it (<= 64 totals: every block sums them itself;
  // block (0, 0) publishes the table the gather kernel and the caller read)
  int out_base = 0;
  for (int b = 0; b < scene; ++b) out_base += min(scene_total[b] + 1, max_voxels);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < batch; ++b) {
      const int t = min(scene_total[b] + 1, max_voxels);
      scene_base[b] = acc;
      voxel_num[b] = t;
      acc += t;
    }
    scene_base[batch] = acc;
  }
  const long long beg = so.off[scene], end = so.off[scene + 1];
  // firsts in earlier tiles of this scene
  int pre = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += blockDim.x) pre += tile_counts[scene * tiles_per_scene + t];
  pre = wave_reduce_sum(pre);
  if (lane_id() == 0) smem[threadIdx.x >> 6] = pre;
  __syncthreads();
  if (threadIdx.x == 0) s_prefix = smem[0] + smem[1] + smem[2] + smem[3];
  __syncthreads();
  const long long base = beg + (long long)blockIdx.x * kTile + threadIdx.x * 4;
  int slot[4];
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = base + j;
    slot[j] = (i < end) ? slot_of_point[i] : -1;
    if (slot[j] >= 0 && (slot[j] & kFirstFlag)) ++cnt;
  }
  int total;
  int rank = block_exclusive_scan(cnt, smem, &total) + s_prefix;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (slot[j] >= 0 && (slot[j] & kFirstFlag)) {
      const int s = slot[j] & ~kFirstFlag;
      const long long i = base + j;
      if (rank < max_voxels) {
        const int vid = out_base + rank;
        vid_of_slot[s] = vid;
        lists[(long long)vid * max_points] = (unsigned)i;
        unsigned cell = (unsigned)(table[s] >> 32) - (unsigned)scene * vol;
        const int cx = cell % g.grid[0];
        cell /= g.grid[0];
        const int cy = cell % g.grid[1];
        const int cz = cell / g.grid[1];
        int* c = coors + (long long)vid * coors_cols;
        if (coors_cols == 4) *c++ = scene;
        c[0] = cz;
        c[1] = cy;
        c[2] = cx;
      } else {
        vid_of_slot[s] = -1;
        if (rank == max_voxels) i_break[scene] = (unsigned)i;  // the point at which the reference breaks
      }
      ++rank;
    }
  }
}

// K4: later points of kept voxels; sorted-list insertion by atomicMin chain.
__global__ void __launch_bounds__(256)
vox_cascade_kernel(SceneOffsets so, const int* __restrict__ slot_of_point, const int* __restrict__ vid_of_slot,
                   const unsigned* __restrict__ i_break, int max_points, unsigned* __restrict__ lists) {
  const int scene = blockIdx.y;
  const long long beg = so.off[scene];
  const long long end = min(so.off[scene + 1], (long long)i_break[scene]);
  for (long long i = beg + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += (long long)gridDim.x * blockDim.x) {
    const int s = slot_of_point[i];
    if (s < 0 || (s & kFirstFlag)) continue;
    const int vid = vid_of_slot[s];
    if (vid < 0) continue;
    unsigned* lst = lists + (long long)vid * max_points;
    unsigned carry = (unsigned)i;
    for (int r = 1; r < max_points; ++r) {
      // (a plain read in front of the atomic -- an entry already below the carry would make it a no-op -- was measured:
      // it adds a dependent round trip per entry, 69 -> 81 us stand-alone; the insert kernel's version of the idea wins)
      const unsigned old = atomicMin(&lst[r], carry);
      if (old == kInf) break;          // landed in an empty entry
      carry = max(old, carry);         // the larger index moves on
    }
  }
}

// K5: one thread per (voxel, feature): copy the <= max_points selected rows, zero pad, mean.
__global__ void __launch_bounds__(256)
vox_gather_kernel(const float* __restrict__ pts, int f, const unsigned* __restrict__ lists, int max_points,
                  const int* __restrict__ scene_base, int batch, float* __restrict__ voxels, int* __restrict__ npv,
                  float* __restrict__ mean) {
  const long long m = scene_base[batch];
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * f) return;
  const long long vid = e / f;
  const int k = (int)(e - vid * f);
  const unsigned* lst = lists + vid * max_points;
  float sum = 0.0f;
  int cnt = 0;
  for (int r = 0; r < max_points; ++r) {
    const unsigned idx = lst[r];
    float v = 0.0f;
    if (idx != kInf) {
      v = pts[(long long)idx * f + k];
      ++cnt;
    }
    voxels[(vid * max_points + r) * f + k] = v;
    sum = __fadd_rn(sum, v);
  }
  if (k == 0) npv[vid] = cnt;
  if (mean) mean[vid * f + k] = __fdiv_rn(sum, (float)cnt);
}

struct HardLayout {
  unsigned tsize;
  int tshift;
  int tiles_per_scene;
  size_t table_b, slot_b, vid_b, lists_b, tiles_b, small_b;
};

HardLayout hard_layout(int64_t n_total, int64_t max_scene_pts, int batch, int max_points, int max_voxels) {
  HardLayout L;
  unsigned t = 1024;
  int lg = 10;
  while ((int64_t)t < 2 * n_total) {
    t <<= 1;
    ++lg;
  }
  L.tsize = t;
  L.tshift = 32 - lg;
  L.tiles_per_scene = (int)std::max<int64_t>(1, ceil_div(max_scene_pts, kTile));
  L.table_b = align_up((size_t)t * 8, 256);
  L.slot_b = align_up((size_t)std::max<int64_t>(n_total, 1) * 4, 256);
  L.vid_b = align_up((size_t)t * 4, 256);
  L.lists_b = align_up((size_t)batch * max_voxels * max_points * 4, 256);
  L.tiles_b = align_up((size_t)batch * L.tiles_per_scene * 4, 256);
  L.small_b = align_up((size_t)(2 * kMaxBatch) * 4, 256) + align_up((size_t)(kMaxBatch + 2) * 4, 256);
  return L;
}

}  // namespace

size_t hash_workspace_bytes(int64_t n_total, int batch, int max_points, int max_voxels) {
  // tiles_per_scene is bounded by the total point count
  HardLayout L = hard_layout(n_total, n_total, batch, max_points, max_voxels);
  return L.table_b + L.slot_b + L.vid_b + L.lists_b + L.tiles_b + L.small_b + 256;
}

int hash_hard_voxelize(const HardArgs& a) {
  hipStream_t stream = a.stream;
  const int batch = a.batch, f = a.f, max_points = a.max_points, max_voxels = a.max_voxels;
  const int64_t n_total = a.n_total, max_scene = a.max_scene;
  const VoxGeom& g = a.g;
  const unsigned long long vol = a.vol;
  const SceneOffsets& so = a.so;
  EFG_CHECK_ARG((long long)batch * max_voxels * max_points < (1ll << 31), "voxel capacity too large");
  EFG_CHECK_ARG(vol * (unsigned long long)batch < 0xffffffffull, "grid volume x batch must be < 2^32-1 cells");
  HardLayout L = hard_layout(n_total, max_scene, batch, max_points, max_voxels);
  Workspace w(a.ws, a.ws_bytes);
  // [table | lists | i_break + scene_total] are contiguous: one 0xff fill makes all of them "empty"
  auto* table = w.take<unsigned long long>(L.tsize);
  unsigned* lists = w.take<unsigned>((size_t)batch * max_voxels * max_points);
  unsigned* cleared_small = w.take<unsigned>(2 * kMaxBatch);
  int* slot_of_point = w.take<int>(std::max<int64_t>(n_total, 1));
  int* vid_of_slot = w.take<int>(L.tsize);
  int* tile_counts = w.take<int>((size_t)batch * L.tiles_per_scene);
  int* small = w.take<int>(kMaxBatch + 2);
  if (!w.ok) {
    set_error("hard_voxelize workspace too small: need %zu bytes, got %zu",
              hash_workspace_bytes(n_total, batch, max_points, max_voxels), a.ws_bytes);
    return EFG_E_WORKSPACE;
  }
  int* scene_base = small;                                   // [batch+1]
  unsigned* i_break = cleared_small;                         // [batch]  0xffffffff = no break
  int* scene_total = reinterpret_cast<int*>(cleared_small + kMaxBatch);  // [batch]  -1 + number of first points
  const dim3 blk(256);
  {
    // (a kernel, not hipMemsetAsync: the memset NODE of a captured call did not take effect on the second replay of the
    // graph on this stack -- scripts/ubench/vox_graph_probe.py; the region is a whole number of 4-byte words)
    const size_t words = (size_t)(reinterpret_cast<char*>(cleared_small + 2 * kMaxBatch) - reinterpret_cast<char*>(table)) / 4;
    hipLaunchKernelGGL(vox_fill_kernel, dim3((unsigned)std::min<size_t>(ceil_div((int64_t)words, 1024), 1024)), blk, 0, stream,
                       reinterpret_cast<unsigned*>(table), words, 0xffffffffu);
  }
  constexpr int precheck = 1;  // (0: every point of a voxel issues its atomicMin -- the losing A/B arm, profiles/r03_vox_precheck*.txt)
  if (n_total > 0) {
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(max_scene, 256), 1), 2048);
    hipLaunchKernelGGL(vox_insert_kernel, dim3(gx, batch), blk, 0, stream, a.points, so, f, g, (unsigned)vol, table,
                       L.tsize - 1, L.tshift, slot_of_point, precheck);
    EFG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(vox_count_kernel, dim3(L.tiles_per_scene, batch), blk, 0, stream, so, table, slot_of_point,
                     tile_counts, L.tiles_per_scene, scene_total);
  EFG_LAUNCH_CHECK();
  hipLaunchKernelGGL(vox_assign_kernel, dim3(L.tiles_per_scene, batch), blk, 0, stream, so, table, slot_of_point,
                     tile_counts, L.tiles_per_scene, scene_total, batch, scene_base, a.voxel_num, max_voxels, max_points,
                     (unsigned)vol, g, vid_of_slot, lists, i_break, a.coors, a.coors_cols);
  EFG_LAUNCH_CHECK();
  if (n_total > 0 && max_points > 1) {
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(max_scene, 256), 1), 2048);
    hipLaunchKernelGGL(vox_cascade_kernel, dim3(gx, batch), blk, 0, stream, so, slot_of_point, vid_of_slot, i_break,
                       max_points, lists);
    EFG_LAUNCH_CHECK();
  }
  // upper bound on rows: min(points, capacity); threads beyond the real count exit early
  const int64_t rows_ub = std::min<int64_t>(n_total, (int64_t)batch * max_voxels);
  if (rows_ub > 0) {
    hipLaunchKernelGGL(vox_gather_kernel, dim3((unsigned)ceil_div(rows_ub * f, 256)), blk, 0, stream, a.points, f,
                       lists, max_points, scene_base, batch, a.voxels, a.npv, a.mean);
    EFG_LAUNCH_CHECK();
  }
  return EFG_OK;
}

}  // namespace efg
