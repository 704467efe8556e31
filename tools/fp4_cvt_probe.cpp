// What v_cvt_scalef32_pk_bf16_fp4 computes (gfx950): element order, scaling by the f32 operand, out-of-range scales.
// hipcc --offload-arch=gfx950 tools/fp4_cvt_probe.cpp -o sgl-kernel-xpu_amd/build/fp4_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, const float* sc, unsigned* out) {
  const unsigned w = in[threadIdx.x];
  const float s = sc[threadIdx.x];
  out[threadIdx.x * 4 + 0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 0));
  out[threadIdx.x * 4 + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 1));
  out[threadIdx.x * 4 + 2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 2));
  out[threadIdx.x * 4 + 3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 3));
}
static float bf(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  unsigned hin[64]; float hsc[64]; unsigned hout[256];
  unsigned scale_bits[8] = {0x3f800000u, 0x40000000u, 0x3f000000u, 0x3fc00000u /*1.5*/, 0x00400000u /*2^-127*/, 0x00800000u /*2^-126*/, 0x7f000000u /*2^127*/, 0x49800000u /*2^20*/};
  for (int i = 0; i < 64; ++i) {
    hin[i] = 0x76543210u ^ ((i & 1) ? 0x88888888u : 0u);
    if (i >= 16) hin[i] = 0xfedcba98u * 0 + (0x10325476u);
    memcpy(&hsc[i], &scale_bits[(i >> 1) & 7], 4);
  }
  unsigned *din, *dout; float* dsc;
  hipMalloc(&din, sizeof hin); hipMalloc(&dsc, sizeof hsc); hipMalloc(&dout, sizeof hout);
  hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice); hipMemcpy(dsc, hsc, sizeof hsc, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dsc, dout);
  hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
  for (int i = 0; i < 18; ++i) {
    printf("in %08x scale %g:", hin[i], hsc[i]);
    for (int b = 0; b < 4; ++b) printf("  [%g %g]", bf(hout[i * 4 + b] & 0xffff), bf(hout[i * 4 + b] >> 16));
    printf("\n");
  }
  return 0;
}
