// Layout of a tile plan (spconv_tiles.hip builds it; the forward / dgrad tile kernel there and the weight-gradient
// kernel of spconv_wgt.hip walk it).
#pragma once
#include <cstddef>

namespace efg {
namespace {

constexpr int kChunkRows = 1024;  // rows sorted together (one workgroup of the plan kernel)

// ---- plan -------------------------------------------------------------------------------------------------
// layout of the plan buffer for (m rows, kvol offsets), n_tiles = round_up(m, 1024) / 16:
//   rows i32 [n_tiles][16] | nb i32 [n_tiles][kvol][16] | vm u32 [n_tiles][32]   (vm[.][31] = active offsets)
//   | pfx1 i32 [n_tiles + 1] | pfx2 i32 [n_tiles / 2 + 1]
// pfxR: exclusive prefix sums of the item counts of the units of R consecutive tiles (stream-K, conv_tile_kernel): a
// unit's items are its active offsets (the union over its tiles); a unit with rows but no offset counts one item, so
// that its rows are still written.
struct PlanView {
  int* rows;
  int* nb;
  unsigned* vm;
  int* pfx1;
  int* pfx2;
  long long n_tiles;
};

__host__ __device__ inline long long plan_tiles(long long m) { return (m + kChunkRows - 1) / kChunkRows * (kChunkRows / 16); }

inline size_t plan_bytes(long long m, int kvol) {
  const long long t = plan_tiles(m);
  return (size_t)t * 16 * 4 + (size_t)t * kvol * 16 * 4 + (size_t)t * 32 * 4 + (size_t)(t + 1) * 4 + (size_t)(t / 2 + 1) * 4;
}

inline PlanView plan_view(void* p, long long m, int kvol) {
  PlanView v;
  v.n_tiles = plan_tiles(m);
  v.rows = static_cast<int*>(p);
  v.nb = v.rows + v.n_tiles * 16;
  v.vm = reinterpret_cast<unsigned*>(v.nb + v.n_tiles * kvol * 16);
  v.pfx1 = reinterpret_cast<int*>(v.vm + v.n_tiles * 32);
  v.pfx2 = v.pfx1 + v.n_tiles + 1;
  return v;
}

}  // namespace
}  // namespace efg
