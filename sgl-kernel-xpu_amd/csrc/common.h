// Shared device/host helpers for the gfx950 kernels (wave64, CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sglk.h"

namespace sglk {

using bf16 = __bf16;
using f16 = _Float16;

constexpr int kWave = 64;

// ---- host-side error plumbing (defined in runtime.hip) ----------------------
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int check_launch(const char* what);
int num_cus();
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function object of the CURRENT device: set it once per
// device (bit d of *done_mask = done on device d; devices >= 64 set it on every launch). Thread-safe.
int set_max_dyn_lds(const void* fn, int bytes, unsigned long long* done_mask, const char* what);

#define SGLK_REQUIRE(cond, ...)                          \
  do {                                                   \
    if (!(cond)) return ::sglk::fail(SGLK_EINVAL, __VA_ARGS__); \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T>
struct DTypeOf;
template <>
struct DTypeOf<float> { static constexpr int v = SGLK_F32; };
template <>
struct DTypeOf<f16> { static constexpr int v = SGLK_F16; };
template <>
struct DTypeOf<bf16> { static constexpr int v = SGLK_BF16; };

// Run fn.template operator()<T>() for a 16/32-bit float dtype code.
#define SGLK_DISPATCH_FLOAT(code, T, ...)                   \
  switch (code) {                                           \
    case SGLK_F32: { using T = float; __VA_ARGS__; break; } \
    case SGLK_F16: { using T = ::sglk::f16; __VA_ARGS__; break; } \
    case SGLK_BF16: { using T = ::sglk::bf16; __VA_ARGS__; break; } \
    default: return ::sglk::fail(SGLK_EUNSUPPORTED, "unsupported float dtype code %d", (int)(code)); \
  }

#define SGLK_DISPATCH_HALF(code, T, ...)                    \
  switch (code) {                                           \
    case SGLK_F16: { using T = ::sglk::f16; __VA_ARGS__; break; } \
    case SGLK_BF16: { using T = ::sglk::bf16; __VA_ARGS__; break; } \
    default: return ::sglk::fail(SGLK_EUNSUPPORTED, "unsupported 16-bit dtype code %d", (int)(code)); \
  }

// ---- fixed-size vectors for coalesced 2..16-byte accesses --------------------
template <typename T, int N>
struct alignas(sizeof(T) * N) Vec {
  T v[N];
  __device__ __forceinline__ T& operator[](int i) { return v[i]; }
  __device__ __forceinline__ const T& operator[](int i) const { return v[i]; }
};

template <typename T, int N>
__device__ __forceinline__ Vec<T, N> load_vec(const T* p) {
  return *reinterpret_cast<const Vec<T, N>*>(p);
}
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const Vec<T, N>& x) {
  *reinterpret_cast<Vec<T, N>*>(p) = x;
}

// ---- wave / block reductions ------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
// reduce over the `width` consecutive lanes a lane belongs to (width = power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_max(float x) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// Sum over all threads of a block of NWAVES waves; `smem` holds >= NWAVES floats.
// Every thread gets the result. Safe to call repeatedly with the same smem.
template <int NWAVES>
__device__ __forceinline__ float block_sum(float x, float* smem) {
  x = wave_sum(x);
  if constexpr (NWAVES == 1) return x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = x;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NWAVES; ++i) t += smem[i];
  return t;
}

// ---- fp8 (OCP e4m3fn) helpers -------------------------------------------------
// Packs 4 floats into 4 e4m3 bytes, round-to-nearest-even (v_cvt_pk_fp8_f32).
// Inputs must already be clamped to [-448, 448].
__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  return (uint32_t)p;
}

}  // namespace sglk
