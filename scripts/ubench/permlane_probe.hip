// What v_permlane32_swap / v_permlane16_swap (gfx950) do to two registers, and the 4 x 4 transpose across the four 16-lane
// groups built from them (csrc/spconv_tiles.hip transpose_pieces):
//   hipcc --offload-arch=gfx950 -o /tmp/pp scripts/ubench/permlane_probe.hip && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void transpose_pieces(float (&v)[4]) {   // lane group g, element e  <->  lane group e, element g
  // v_permlane32_swap a, b: a's lanes 32..63 <-> b's lanes 0..31; v_permlane16_swap a, b: a's odd 16-lane rows <-> b's even
  // rows (gfx950; scripts/ubench/permlane_probe.hip prints both).  Inline asm with both registers read-write: chained through
  // __builtin_amdgcn_permlane*_swap, this compiler (ROCm 7.2) returns the FIRST result for both halves of the second pair
  // (the probe's "128 of 256 wrong"); the s_nop covers the VALU-write -> permlane-read hazard the compiler would otherwise pad.
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
__global__ void k(unsigned* o, float* t) {
  const unsigned l = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
  o[l] = a[0];
  o[64 + l] = a[1];
  o[128 + l] = b[0];
  o[192 + l] = b[1];
  float v[4];
  for (int e = 0; e < 4; ++e) v[e] = 10.0f * (l >> 4) + e + 0.01f * (l & 15);   // group g, element e, row i
  transpose_pieces(v);
  for (int e = 0; e < 4; ++e) t[l * 4 + e] = v[e];
}
int main() {
  unsigned* d;
  float* t;
  unsigned h[256];
  float ht[256];
  hipMalloc(&d, sizeof(h));
  hipMalloc(&t, sizeof(ht));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane32_swap(x = lane, y = 100 + lane)[0]", "permlane32_swap[1]", "permlane16_swap(x, y)[0]", "permlane16_swap[1]"};
  for (int r = 0; r < 4; ++r)
    printf("%s: lanes 0, 16, 32, 48 -> %u %u %u %u\n", names[r], h[r * 64], h[r * 64 + 16], h[r * 64 + 32], h[r * 64 + 48]);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      const float want = 10.0f * e + (l >> 4) + 0.01f * (l & 15);   // element e of group g after = element g of group e before
      if (ht[l * 4 + e] != want) ++bad;
    }
  printf("transpose of v[e] = 10 g + e + 0.01 i: lane 17 holds %.2f %.2f %.2f %.2f (want 1.01 11.01 21.01 31.01); %d of 256 wrong\n",
         ht[17 * 4], ht[17 * 4 + 1], ht[17 * 4 + 2], ht[17 * 4 + 3], bad);
  return 0;
}
