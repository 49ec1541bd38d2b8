// micro-benchmark: LDS update throughput on gfx950 (ds_add_f32 vs ds_add_u32 vs read-modify-write)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, int stride) {
  __shared__ float buf[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = 0.f;
  __syncthreads();
  unsigned a = (threadIdx.x * stride) & 16383;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      unsigned idx = (a + u * 1024 + it * 64) & 16383;
      if (MODE == 0) atomicAdd(&buf[idx], 1.0f);
      else if (MODE == 1) atomicAdd((unsigned*)&buf[idx], 1u);
      else if (MODE == 2) { float v = buf[idx]; buf[idx] = v + 1.0f; }
      else if (MODE == 3) { buf[idx] = (float)it; }
      else if (MODE == 5) { float r = atomicAdd(&buf[idx], 1.0f); if (r == 12345.f) buf[0] = r; }
      else if (MODE == 6) { atomicAdd((double*)&buf[idx & ~1u], 1.0); }
      else if (MODE == 7) { __hip_atomic_fetch_add(&buf[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      else if (MODE == 8) { atomicMax((int*)&buf[idx], it); }
      else if (MODE == 9) { atomicAdd((unsigned long long*)&buf[idx & ~1u], 1ull); }
      else if (MODE == 4) { float4* p = (float4*)&buf[idx & ~3u]; float4 v = *p; v.x += 1; v.y += 1; v.z += 1; v.w += 1; *p = v; }
    }
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < 16384; i += 256) s += buf[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int stride) {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  int iters = 200, blocks = 2048;
  k<MODE><<<blocks, 256>>>(out, iters, stride); hipDeviceSynchronize();
  hipEventRecord(s); k<MODE><<<blocks, 256>>>(out, iters, stride); hipEventRecord(e); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, s, e);
  double ops = (double)blocks * 256 * iters * 16 * (MODE == 4 ? 4 : 1);
  printf("%-28s stride %2d: %8.3f ms  %8.1f G lane-ops/s  (%.2f lane-ops/clk/CU @2.4GHz)\n", name, stride, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
  hipFree(out);
}
int main() {
  for (int stride : {1, 4}) {
    run<0>("ds_add_f32", stride);
    run<1>("ds_add_u32", stride);
    run<2>("ds_read+add+ds_write b32", stride);
    run<3>("ds_write_b32", stride);
    run<4>("rmw b128 (4 floats/lane)", stride);
    run<5>("ds_add_rtn_f32", stride);
    run<6>("ds_add_f64", stride);
    run<7>("hip_atomic wg-scope f32", stride);
    run<8>("ds_max_i32", stride);
    run<9>("ds_add_u64", stride);
  }
  return 0;
}
