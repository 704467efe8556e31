// bench.py's measured MFMA ceiling (NOT part of the product library): a register-only stream of the MX fp8 MFMA the headline
// GEMM uses (v_mfma_scale_f32_32x32x64_f8f6f4, unit scales), random e4m3 operands held in registers, no memory traffic in
// the loop, 512-thread workgroups = two waves per SIMD as in the GEMM, one workgroup per CU. What this loop reaches on
// the device bench.py runs on is what the matrix pipes deliver at the clock the part holds under them (MI355X_MICROARCH.md,
// "DVFS give-back"): the denominator next to the nominal 5 PFLOP/s.
//   build: sgl-kernel-xpu_amd/build.py -> sgl-kernel-xpu_amd/build/libsglk_ceiling.so (ctypes from bench.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_ceiling_kernel(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  v8i_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = src[(threadIdx.x * 8 + j + i * 4096) & 16383];
      b[i][j] = src[(threadIdx.x * 8 + j + i * 4096 + 777) & 16383];
    }
  v16f_t acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b[n], a[u + 2 * (it & 1)], acc[n], 0, 0, 0, 127, 0, 127);
  }
  float s = 0;
  for (int n = 0; n < 4; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// src: 16384 random int32 (e4m3 bytes without NaN codes), dst: blocks * 512 floats. FLOP per launch = flop_per_launch().
extern "C" __attribute__((visibility("default"))) int sglk_bench_mfma_ceiling(void* stream, const void* src, void* dst,
                                                                              int blocks, int iters) {
  mfma_ceiling_kernel<<<blocks, 512, 0, (hipStream_t)stream>>>((const int*)src, (float*)dst, iters);
  return (int)hipGetLastError();
}
// 8 waves x 8 MFMAs per iteration x 2 * 32 * 32 * 64 FLOP
extern "C" __attribute__((visibility("default"))) double sglk_bench_mfma_ceiling_flop(int blocks, int iters) {
  return (double)blocks * 8.0 * 8.0 * iters * 2.0 * 32 * 32 * 64;
}

// The same for the bf16 kernels (flash_mla_decode, fwd prefill, the MoE tile pipeline): v_mfma_f32_32x32x16_bf16 on random
// bf16 operands (exponent bits masked so that nothing is NaN / Inf), `waves` waves per workgroup (4 = one per SIMD as in
// mla_rows128z_kernel, 8 = two as in the prefill / MoE kernels), one workgroup per CU.
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma_ceiling_bf16_kernel(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  v4i_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = src[(threadIdx.x * 4 + j + i * 4096) & 16383] & 0xBFFFBFFF;
      b[i][j] = src[(threadIdx.x * 4 + j + i * 4096 + 777) & 16383] & 0xBFFFBFFF;
    }
  v16f_t acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf_t, b[n]),
                                                         __builtin_bit_cast(v8bf_t, a[u + 2 * (it & 1)]), acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < 4; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" __attribute__((visibility("default"))) int sglk_bench_mfma_ceiling_bf16(void* stream, const void* src, void* dst,
                                                                                   int blocks, int waves, int iters) {
  mfma_ceiling_bf16_kernel<<<blocks, 64 * waves, 0, (hipStream_t)stream>>>((const int*)src, (float*)dst, iters);
  return (int)hipGetLastError();
}
// waves x 8 MFMAs per iteration x 2 * 32 * 32 * 16 FLOP
extern "C" __attribute__((visibility("default"))) double sglk_bench_mfma_ceiling_bf16_flop(int blocks, int waves, int iters) {
  return (double)blocks * waves * 8.0 * iters * 2.0 * 32 * 32 * 16;
}

// Shape probe (MI355X_MICROARCH.md, DVFS give-back item 7: the clock the part holds can depend on the MFMA shape): the same
// FLOP per iteration and the same 64 x 64 output tile per wave as the kernels above, on v_mfma_f32_16x16x32_bf16 and on
// v_mfma_scale_f32_16x16x128_f8f6f4. lds = 1: every A / B fragment is re-read from LDS by ds_read_b128 in front of its use.
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <int LDS>
__global__ __launch_bounds__(512) void mfma_ceiling_bf16_16_kernel(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  __shared__ v4i_t frag[8 * 512];
  v4i_t a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) {
      a[i][j] = src[(threadIdx.x * 4 + j + i * 4096) & 16383] & 0xBFFFBFFF;
      b[i][j] = src[(threadIdx.x * 4 + j + i * 4096 + 777) & 16383] & 0xBFFFBFFF;
    }
    frag[i * 512 + threadIdx.x] = a[i];
    frag[(4 + i) * 512 + threadIdx.x] = b[i];
  }
  __syncthreads();
  v4f_t acc[16] = {};
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = frag[i * 512 + ((threadIdx.x + it) & 511)];
        b[i] = frag[(4 + i) * 512 + ((threadIdx.x + it) & 511)];
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, a[m]), __builtin_bit_cast(v8bf_t, b[n]),
                                                                 acc[m * 4 + n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < 16; ++n)
    for (int r = 0; r < 4; ++r) s += acc[n][r];
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int LDS>
__global__ __launch_bounds__(512) void mfma_ceiling_bf16_32_kernel(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  __shared__ v4i_t frag[8 * 512];
  v4i_t a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) {
      a[i][j] = src[(threadIdx.x * 4 + j + i * 4096) & 16383] & 0xBFFFBFFF;
      b[i][j] = src[(threadIdx.x * 4 + j + i * 4096 + 777) & 16383] & 0xBFFFBFFF;
    }
    frag[i * 512 + threadIdx.x] = a[i];
    frag[(4 + i) * 512 + threadIdx.x] = b[i];
  }
  __syncthreads();
  v16f_t acc[4] = {};
  for (int it = 0; it < iters; ++it) {
    if (LDS) {  // (two k-steps of a 64 x 64 tile: 2 + 2 fragments each, eight reads per iteration as in the 16-wide loop)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = frag[i * 512 + ((threadIdx.x + it) & 511)];
        b[i] = frag[(4 + i) * 512 + ((threadIdx.x + it) & 511)];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf_t, b[(n & 1) + 2 * u]),
                                                         __builtin_bit_cast(v8bf_t, a[(n >> 1) + 2 * u]), acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < 4; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// shape: 0 = 32x32x16, 1 = 16x16x32; the FLOP count is sglk_bench_mfma_ceiling_bf16_flop's
extern "C" __attribute__((visibility("default"))) int sglk_bench_mfma_shape_bf16(void* stream, const void* src, void* dst, int blocks,
                                                                                 int waves, int iters, int shape, int lds) {
  hipStream_t st = (hipStream_t)stream;
  if (shape == 0 && !lds) mfma_ceiling_bf16_32_kernel<0><<<blocks, 64 * waves, 0, st>>>((const int*)src, (float*)dst, iters);
  else if (shape == 0) mfma_ceiling_bf16_32_kernel<1><<<blocks, 64 * waves, 0, st>>>((const int*)src, (float*)dst, iters);
  else if (!lds) mfma_ceiling_bf16_16_kernel<0><<<blocks, 64 * waves, 0, st>>>((const int*)src, (float*)dst, iters);
  else mfma_ceiling_bf16_16_kernel<1><<<blocks, 64 * waves, 0, st>>>((const int*)src, (float*)dst, iters);
  return (int)hipGetLastError();
}

// fp8 MX: v_mfma_scale_f32_16x16x128_f8f6f4, 16 accumulators of 16 x 16, FLOP per iteration = the 32x32x64 loop's
__global__ __launch_bounds__(512) void mfma_ceiling_fp8_16_kernel(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  v8i_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = src[(threadIdx.x * 8 + j + i * 4096) & 16383];
      b[i][j] = src[(threadIdx.x * 8 + j + i * 4096 + 777) & 16383];
    }
  v4f_t acc[16] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[m * 4 + n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m], b[n], acc[m * 4 + n], 0, 0, 0, 127, 0, 127);
  }
  float s = 0;
  for (int n = 0; n < 16; ++n)
    for (int r = 0; r < 4; ++r) s += acc[n][r];
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 512-thread workgroups; FLOP per launch = sglk_bench_mfma_ceiling_flop's (16 x 2 * 16 * 16 * 128 = 8 x 2 * 32 * 32 * 64)
extern "C" __attribute__((visibility("default"))) int sglk_bench_mfma_shape_fp8_16(void* stream, const void* src, void* dst, int blocks,
                                                                                   int iters) {
  mfma_ceiling_fp8_16_kernel<<<blocks, 512, 0, (hipStream_t)stream>>>((const int*)src, (float*)dst, iters);
  return (int)hipGetLastError();
}
