// micro-benchmark: what keeps v_mfma_f32_16x16x4_f32 from issuing back to back on gfx950?
//   MODE 0: 128 MFMAs per trip on 8 accumulators, operands in registers               (the pipe's ceiling)
//   MODE 1: + the A operands of every 16 MFMAs read from LDS (8 ds_read_b32), as the sparse-conv step does
//   MODE 2: + 16 B operand float4 loads per trip from a 16 KB L2-resident block
//   MODE 3: + 32 four-byte gathers and 32 LDS writes per trip (the whole step, no neighbour table)
//   MODE 4: MODE 3 with the gather addresses read from an LDS table (32 ds_read_b32 + 32 adds) and the stash masked
//   MODE 6: the A operands loaded straight in fragment layout -- lane (m, kk) loads 16 bytes of ITS row (16 rows x 64 bytes per
//           instruction; row offset in a register of the lane: no LDS, no table) -- 8 loads per trip; weights and rows as MODE 5
//   MODE 5: MODE 4 with the weights drawn from 448 KB (a 64-channel layer's) and the rows from 12 MB, pseudo-random
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_issue scripts/ubench/mfma_f32_issue.hip ; run: waves per SIMD as argv[1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ w, const float* __restrict__ x, float* out, int trips) {
  __shared__ float at[4][32 * 66];
  __shared__ int nbt[4][1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* a0 = at[wv];
  for (int i = lane; i < 32 * 66; i += 64) a0[i] = (float)(i & 7);
  int* nb = nbt[wv];
  for (int i = lane; i < 1024; i += 64) nb[i] = (int)(((unsigned)(i * 2654435761u + blockIdx.x * 40503u) >> 8) % (MODE >= 5 ? 45760u : 8192u)) * 256;
  __builtin_amdgcn_wave_barrier();
  f32x4 acc[2][4];
  for (int s = 0; s < 2; ++s)
    for (int t = 0; t < 4; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float af[8];
  for (int q = 0; q < 8; ++q) af[q] = (float)(lane + q);
  float4 bq[4];
  for (int t = 0; t < 4; ++t) bq[t] = make_float4(1.f + t, 2.f, 3.f, 4.f);
  const int m = lane & 15, kk = lane >> 4;
  float pre[32];
  for (int j = 0; j < 32; ++j) pre[j] = 0.f;
  if (MODE == 6) {
    // per-lane row offsets of the current and the next trip (one 4-byte load per sub-tile and trip in the real kernel)
    f32x4 an[8];
    auto rows_of = [&](int it, unsigned (&ro)[2]) {
      for (int s = 0; s < 2; ++s) ro[s] = (unsigned)nb[((it * 7) & 31) * 32 + s * 16 + m] + kk * 16u;
    };
    auto load_a = [&](const unsigned (&ro)[2], f32x4 (&a)[8]) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[s * 4 + i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(x) + ro[s] + i * 64);
    };
    unsigned ro[2];
    rows_of(0, ro);
    load_a(ro, an);
    for (int it = 0; it < trips; ++it) {
      f32x4 ac[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) ac[q] = an[q];
      rows_of(it + 1, ro);
      load_a(ro, an);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          bq[t] = *reinterpret_cast<const float4*>(w + (((it * 11 + blockIdx.x) % 27) * 4 + i) * 1024 + t * 256 + lane * 4);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float bb = q == 0 ? bq[t].x : (q == 1 ? bq[t].y : (q == 2 ? bq[t].z : bq[t].w));
              acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[s * 4 + i][q], bb, acc[s][t], 0, 0, 0);
            }
      }
    }
  } else
  for (int it = 0; it < trips; ++it) {
    unsigned msk = 0xffffffffu;
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 32; ++j) pre[j] = x[(size_t)((blockIdx.x * 37 + it * 11 + j * 5) & 8191) * 64 + lane];
    }
    if (MODE >= 4) {
      const int col = (it * 7) & 31;
      msk = __builtin_amdgcn_readfirstlane(nb[col] | 0xfffffff0);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const unsigned off = (unsigned)nb[col * 32 + j] + lane * 4u;
        pre[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + off);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE >= 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          bq[t] = *reinterpret_cast<const float4*>(w + ((MODE >= 5 ? ((it * 11 + blockIdx.x) % 27) : (it & 3)) * 4 + i) * 1024 + t * 256 + lane * 4);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (MODE >= 1) {
          const float* ap = a0 + (s * 16 + m) * 66 + i * 16 + kk;
          af[s * 4 + 0] = ap[0], af[s * 4 + 1] = ap[4], af[s * 4 + 2] = ap[8], af[s * 4 + 3] = ap[12];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float bb = q == 0 ? bq[t].x : (q == 1 ? bq[t].y : (q == 2 ? bq[t].z : bq[t].w));
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + q], bb, acc[s][t], 0, 0, 0);
          }
      }
    }
    if (MODE >= 3) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 32; ++j) a0[j * 66 + lane] = MODE >= 4 ? __int_as_float(__float_as_int(pre[j]) & -(int)((msk >> j) & 1u)) : pre[j];
      __builtin_amdgcn_wave_barrier();
    }
  }
  float s = 0;
  for (int a = 0; a < 2; ++a)
    for (int t = 0; t < 4; ++t) s += acc[a][t][0] + acc[a][t][1] + acc[a][t][2] + acc[a][t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(int wgs, int trips, const float* w, const float* x, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<wgs, 256>>>(w, x, out, trips);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE><<<wgs, 256>>>(w, x, out, trips);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 5.0 * wgs * 4 * (double)trips * 128 * 2048;
  printf("mode %d: %d workgroups x %d trips: %.1f us per launch, %.1f TFLOP/s (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", MODE, wgs, trips,
         ms * 1e3 / 5, flops / (ms * 1e-3) / 1e12, ms * 1e-3 / 5 * 2.4e9 / ((double)wgs * 4 / 1024 * trips * 128));
}
int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 1, trips = argc > 2 ? atoi(argv[2]) : 200;
  float *w, *x, *out;
  hipMalloc(&w, 27 * 16 * 1024 * 4);
  hipMalloc(&x, 45760 * 64 * 4);
  hipMalloc(&out, 1024 * 8 * 256 * 4);
  hipMemset(w, 0, 27 * 16 * 1024 * 4);
  hipMemset(x, 0, 45760 * 64 * 4);
  const int wgs = 256 * waves;
  run<0>(wgs, trips, w, x, out);
  run<1>(wgs, trips, w, x, out);
  run<2>(wgs, trips, w, x, out);
  run<3>(wgs, trips, w, x, out);
  run<4>(wgs, trips, w, x, out);
  run<5>(wgs, trips, w, x, out);
  run<6>(wgs, trips, w, x, out);
  return 0;
}
