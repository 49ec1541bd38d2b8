// Point-cloud voxelization for gfx950 (wave64): entry points.
//
// Replaces efg::dynamic_voxelize / efg::hard_voxelize
// (reference: efg/operators/src/voxelize/voxelization.h:51-83; CPU semantics
// voxelization_cpu.cpp:7-99; the CUDA version voxelization_cuda.cu:100-174 is an O(N^2)
// predecessor scan plus a <<<1,1>>> serial pass).
//
// hard voxelization has two implementations behind one entry point:
//   voxelize_bins.hip  (default) points binned into BEV supercells, per-bin cell tables in LDS;
//   voxelize_hash.hip  (EFG_VOX_IMPL=hash, and grids that cannot be binned) one global hash table.
// Both reproduce the serial first-come loop bit for bit.
#include "voxelize_common.h"

namespace efg {
namespace {

__global__ void __launch_bounds__(256) dynamic_voxelize_kernel(const float* __restrict__ pts, long long n, int f,
                                                                VoxGeom g, int* __restrict__ coors) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    int cx, cy, cz;
    const bool ok = point_cell(pts + i * f, g, cx, cy, cz);
    coors[i * 3 + 0] = ok ? cz : -1;
    coors[i * 3 + 1] = ok ? cy : -1;
    coors[i * 3 + 2] = ok ? cx : -1;
  }
}

int make_geom(const float* vs, const float* cr, VoxGeom* g, unsigned long long* vol) {
  for (int i = 0; i < 3; ++i) {
    EFG_CHECK_ARG(vs[i] > 0.0f, "voxel_size[%d] must be positive", i);
    g->vs[i] = vs[i];
    g->rmin[i] = cr[i];
    // grid = round((max - min) / vs) in fp32, voxelization_cpu.cpp:119-122
    g->grid[i] = (int)roundf((cr[3 + i] - cr[i]) / vs[i]);
    EFG_CHECK_ARG(g->grid[i] > 0, "empty grid on axis %d", i);
  }
  *vol = (unsigned long long)g->grid[0] * g->grid[1] * g->grid[2];
  return EFG_OK;
}

// 1: hash, 0: binned.  EFG_VOX_IMPL=hash|bins forces one (read per call -- a getenv is nothing next to six launches --
// so tests flip it in-process); otherwise by size: the binned path has a fixed cost (seven launches, a scan over every
// supercell of the grid) that a small cloud does not amortise -- 16k points: 44 us against 31 us for the hash path,
// 2 x 180k: 80 against 94, 8 x 180k: 181 against 288 (profiles/r04_vox_times.txt).
constexpr int64_t kBinsMinPoints = 65536;

// The binned path keeps one counter row per XCD and updates it with atomics that stay in the L2 of the XCD the workgroup
// runs on (voxelize_bins.hip K1: the row is picked by the XCC_ID hardware register, 8 rows).  That is only a race-free
// protocol on a part whose XCC_ID values are distinct per L2 and fit the 8 rows: gfx950 (MI350X / MI355X, at most 8 XCDs
// per device in every partition mode).  Queried once per device; anything else takes the hash path (agent-scope atomics).
bool bins_supported_here() {
  static std::atomic<unsigned long long> known{0}, ok{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  const unsigned long long bit = 1ull << dev;
  if (!(known.load(std::memory_order_acquire) & bit)) {
    hipDeviceProp_t prop;
    bool good = hipGetDeviceProperties(&prop, dev) == hipSuccess && !strncmp(prop.gcnArchName, "gfx950", 6) &&
                prop.multiProcessorCount <= 8 * 32;   // (8 XCDs x 32 CUs: more CUs than that is not the part this was built for)
    if (good) ok.fetch_or(bit, std::memory_order_relaxed);
    known.fetch_or(bit, std::memory_order_release);
  }
  return (ok.load(std::memory_order_relaxed) & bit) != 0;
}

int pick_impl(int64_t n_total) {
  const char* e = getenv("EFG_VOX_IMPL");
  if (e && !strcmp(e, "hash")) return 1;
  if (!bins_supported_here()) return 1;
  if (e && !strcmp(e, "bins")) return 0;
  return n_total < kBinsMinPoints ? 1 : 0;
}

}  // namespace
}  // namespace efg

using namespace efg;

extern "C" int efg_dynamic_voxelize_f32(const float* points, int64_t n, int f, const float* vs, const float* cr,
                                        int32_t* coors, void* stream) {
  EFG_CHECK_ARG(f >= 3, "points need >= 3 features, got %d", f);
  EFG_CHECK_ARG(n >= 0, "negative point count");
  VoxGeom g;
  unsigned long long vol;
  if (int rc = make_geom(vs, cr, &g, &vol)) return rc;
  if (n == 0) return EFG_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 256 * 8);
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, points,
                     (long long)n, f, g, coors);
  EFG_LAUNCH_CHECK();
  return EFG_OK;
}

extern "C" size_t efg_hard_voxelize_workspace_bytes(int64_t n_total, int batch, int f, int max_points, int max_voxels,
                                                    const float* vs, const float* cr) {
  if (n_total < 0 || batch < 1 || batch > kMaxBatch || f < 3 || max_points < 1 || max_voxels < 1) return 0;
  VoxGeom g;
  unsigned long long vol;
  if (make_geom(vs, cr, &g, &vol)) return 0;
  // enough for either implementation, so that EFG_VOX_IMPL can be flipped without re-sizing
  return std::max(hash_workspace_bytes(n_total, batch, max_points, max_voxels),
                  bins_workspace_bytes(n_total, batch, f, g));
}

extern "C" int efg_hard_voxelize_f32(const float* points, const int64_t* offs, int batch, int f, const float* vs,
                                     const float* cr, int max_points, int max_voxels, float* voxels, int32_t* coors,
                                     int coors_cols, int32_t* npv, int32_t* voxel_num, float* mean, void* ws,
                                     size_t ws_bytes, void* stream_) {
  HardArgs a;
  a.stream = (hipStream_t)stream_;
  EFG_CHECK_ARG(batch >= 1 && batch <= kMaxBatch, "batch must be in [1,%d], got %d", kMaxBatch, batch);
  EFG_CHECK_ARG(f >= 3, "points need >= 3 features, got %d", f);
  EFG_CHECK_ARG(max_points >= 1 && max_voxels >= 1,
                "max_points/max_voxels must be >= 1 (the -1 'uncapped' modes are routed to dynamic_voxelize by "
                "efg/operators/voxelize.py:34-37)");
  EFG_CHECK_ARG(coors_cols == 3 || coors_cols == 4, "coors_cols must be 3 or 4");
  if (int rc = make_geom(vs, cr, &a.g, &a.vol)) return rc;
  a.max_scene = 0;
  for (int b = 0; b <= batch; ++b) {
    a.so.off[b] = offs[b];
    if (b) {
      EFG_CHECK_ARG(offs[b] >= offs[b - 1], "point_offsets must be non-decreasing");
      a.max_scene = std::max<int64_t>(a.max_scene, offs[b] - offs[b - 1]);
    }
  }
  EFG_CHECK_ARG(offs[0] == 0, "point_offsets[0] must be 0");
  a.n_total = offs[batch];
  EFG_CHECK_ARG(a.n_total < (1ll << 29), "too many points");
  a.points = points;
  a.batch = batch;
  a.f = f;
  a.max_points = max_points;
  a.max_voxels = max_voxels;
  a.coors_cols = coors_cols;
  a.voxels = voxels;
  a.coors = coors;
  a.npv = npv;
  a.voxel_num = voxel_num;
  a.mean = mean;
  a.ws = ws;
  a.ws_bytes = ws_bytes;
  if (pick_impl(a.n_total) == 1 || bins_workspace_bytes(a.n_total, batch, f, a.g) == 0) return hash_hard_voxelize(a);
  return bins_hard_voxelize(a);
}

extern "C" size_t efg_hard_voxelize_debug_timeline(void* device_buf_u64) {
  bins_set_debug_timeline(static_cast<unsigned long long*>(device_buf_u64));
  return bins_debug_timeline_words();
}
