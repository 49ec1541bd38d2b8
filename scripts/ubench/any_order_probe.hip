// Does hipExtAnyOrderLaunch drop the AQL barrier bit on gfx950 (ROCm 7.2)?  hip_ext.h says "not supported on AMD GFX9xx
// boards" for the module-launch form.  Two kernels that each occupy HALF of the chip for ~200 us, launched back to back on
// ONE stream: in order they take 2 x T; if the second packet carries no barrier bit they run side by side in ~T.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/any_order_probe.hip -o scripts/ubench/any_order_probe && scripts/ubench/any_order_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void __launch_bounds__(256) spin_kernel(float* out, long long cycles) {
  const long long t0 = wall_clock64();   // 100 MHz constant clock
  float v = (float)threadIdx.x;
  while (wall_clock64() - t0 < cycles) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) out[blockIdx.x] = v;
}

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                          \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main() {
  float* out = nullptr;
  CK(hipMalloc(&out, 4096 * sizeof(float)));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long long cycles = 20000;   // 200 us at 100 MHz
  for (int wgs : {128, 256, 1024}) {
    for (int mode = 0; mode < 3; ++mode) {   // 0: one kernel, 1: two in order, 2: second with hipExtAnyOrderLaunch
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st, out, cycles);
        if (mode == 1) hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st, out, cycles);
        if (mode == 2) hipExtLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, out, cycles);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("workgroups %4d  %-28s %8.1f us\n", wgs, mode == 0 ? "one kernel" : mode == 1 ? "two, in order" : "two, second any-order", best * 1e3f);
    }
  }
  return 0;
}
