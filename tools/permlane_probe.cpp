// probe (inline asm; the builtins of this compiler return the first result twice): semantics of v_permlane16_swap / v_permlane32_swap on gfx950 (xor-16 / xor-32 max reduction)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(float* o) {
  float x = (float)((threadIdx.x * 37) % 64);
  float a0 = x, a1 = x;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
  o[threadIdx.x] = a0;
  o[64 + threadIdx.x] = a1;
  float c0 = fmaxf(a0, a1), c1 = c0;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c0), "+v"(c1));
  o[128 + threadIdx.x] = fmaxf(c0, c1);
  o[192 + threadIdx.x] = x;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int part = 0; part < 4; ++part) { printf("part %d:", part); for (int i = 0; i < 64; ++i) printf(" %g", h[part * 64 + i]); printf("\n"); }
  // expected part 2: for lane l, max over lanes {l, l^16, l^32, l^48} of x
  int bad = 0;
  for (int l = 0; l < 64; ++l) { float e = 0; for (int m : {0, 16, 32, 48}) e = fmaxf(e, h[192 + (l ^ m)]); if (e != h[128 + l]) ++bad; }
  printf("bad %d\n", bad);
  return 0;
}
