// micro-benchmark: bytes per clock a CU gets out of the L2 with 16-byte loads (the weight-fragment loads of the sparse
// convolution), by working-set size and by whether all CUs read the SAME bytes (one layer's weights) or their own.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/l2_fill_rate scripts/ubench/l2_fill_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int WIDTH>
__global__ void __launch_bounds__(256) k(const float* __restrict__ w, float* out, int trips, unsigned region_floats, unsigned per_wg_off) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* base = w + (size_t)blockIdx.x * per_wg_off;
  float s = 0;
  unsigned pos = (wv * 4099u + blockIdx.x * 8191u) * 1024u;   // every wave walks its own order of 4 KB blocks
  for (int it = 0; it < trips; ++it) {
    float4 v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const unsigned o = (pos + t * 256u) % region_floats;   // 16 x 1 KB = one 16 KB weight block per trip
      if (WIDTH == 16) v[t] = *reinterpret_cast<const float4*>(base + o + lane * 4);
      else v[t].x = base[o + lane], v[t].y = base[o + 64 + lane], v[t].z = base[o + 128 + lane], v[t].w = base[o + 192 + lane];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) s += v[t].x + v[t].y + v[t].z + v[t].w;
    pos += 4096u * 7u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int WIDTH>
void run(int waves, int trips, const float* w, float* out, unsigned region_kb, bool shared) {
  const int wgs = 256 * waves;
  const unsigned region_floats = region_kb * 256u, off = shared ? 0u : region_floats;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<WIDTH><<<wgs, 256>>>(w, out, trips, region_floats, off);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<WIDTH><<<wgs, 256>>>(w, out, trips, region_floats, off);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 5.0 * wgs * 4 * (double)trips * 16384;
  printf("  %2d-byte loads, %5u KB %s, %d waves/SIMD: %7.2f TB/s = %5.1f B/clk/CU\n", WIDTH, region_kb, shared ? "shared by all CUs" : "per workgroup     ",
         waves, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.4e9);
}
int main(int argc, char** argv) {
  const int trips = argc > 1 ? atoi(argv[1]) : 400;
  float *w, *out;
  const size_t total = (size_t)1024 * 64 * 1024 * 4;   // 256 MB
  (void)hipMalloc(&w, total);
  (void)hipMalloc(&out, 1024 * 8 * 256 * 4);
  (void)hipMemset(w, 0, total);
  for (int waves : {1, 2, 4}) {
    for (unsigned kb : {64u, 448u, 1792u, 7168u}) run<16>(waves, trips, w, out, kb, true);
    run<16>(waves, trips, w, out, 64, false);
    run<4>(waves, trips, w, out, 448, true);
  }
  return 0;
}
