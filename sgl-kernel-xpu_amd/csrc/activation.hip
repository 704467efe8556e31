// silu_and_mul / gelu_tanh_and_mul / gelu_and_mul for gfx950.
//
// Replaces reference src/sycl/TripleOps.cpp:29-80 (functors), :140-235 (hosts).
// Arithmetic kept (fp32 opmath, one rounding to T at the end):
//   silu      : (a / (1 + expf(-a))) * b                        (TripleOps.cpp:31)
//   gelu_tanh : (0.5*a*(1 + tanh(kBeta*(a + 0.044715*a^3)))) * b (TripleOps.cpp:38-43)
//   gelu      : (a*0.5*(1 + erf(a*M_SQRT1_2))) * b              (TripleOps.cpp:49-51)
// Design: pure HBM stream (2 reads + 1 write per output element). The token
// and column are recovered from a flat vector index so that any (tokens, d)
// fills the chip; every lane moves 16 bytes per access.
#include <math.h>

#include "common.h"

namespace sglk {
namespace {

template <int ACT>
__device__ __forceinline__ float act_mul(float a, float b) {
  if constexpr (ACT == SGLK_ACT_SILU) {
    return (a / (1.0f + expf(-a))) * b;
  } else if constexpr (ACT == SGLK_ACT_GELU_TANH) {
    const float kBeta = (float)(M_SQRT2 * M_2_SQRTPI * 0.5);
    const float kKappa = 0.044715f;
    const float cube = a * a * a;
    const float inner = kBeta * (a + kKappa * cube);
    return (0.5f * a * (1.0f + tanhf(inner))) * b;
  } else {
    return (a * 0.5f * (1.0f + erff(a * (float)M_SQRT1_2))) * b;
  }
}

template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(256) void act_and_mul_kernel(T* __restrict__ out,
                                                          const T* __restrict__ x,
                                                          int64_t total_vecs, int dvec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_vecs; idx += stride) {
    const int64_t tok = idx / dvec;
    const int j = (int)(idx - tok * dvec);
    const T* row = x + tok * (int64_t)dvec * 2 * VEC;
    Vec<T, VEC> a = load_vec<T, VEC>(row + (int64_t)j * VEC);
    Vec<T, VEC> b = load_vec<T, VEC>(row + ((int64_t)dvec + j) * VEC);
    Vec<T, VEC> y;
#pragma unroll
    for (int e = 0; e < VEC; ++e) y[e] = (T)act_mul<ACT>((float)a[e], (float)b[e]);
    store_vec<T, VEC>(out + idx * VEC, y);
  }
}

template <typename T, int VEC>
static int launch(hipStream_t st, T* out, const T* x, int64_t tokens, int64_t d, int act) {
  const int64_t dvec = d / VEC;
  const int64_t total = tokens * dvec;
  const int64_t want = cdiv(total, 256);
  const unsigned blocks = (unsigned)(want < 256 * 16 ? want : 256 * 16);
  switch (act) {
    case SGLK_ACT_SILU:
      act_and_mul_kernel<T, VEC, SGLK_ACT_SILU><<<blocks, 256, 0, st>>>(out, x, total, (int)dvec);
      break;
    case SGLK_ACT_GELU_TANH:
      act_and_mul_kernel<T, VEC, SGLK_ACT_GELU_TANH><<<blocks, 256, 0, st>>>(out, x, total, (int)dvec);
      break;
    case SGLK_ACT_GELU:
      act_and_mul_kernel<T, VEC, SGLK_ACT_GELU><<<blocks, 256, 0, st>>>(out, x, total, (int)dvec);
      break;
    default:
      return fail(SGLK_EINVAL, "act_and_mul: unknown activation %d", act);
  }
  return check_launch("act_and_mul");
}

// ---- silu_and_mul_clamp: reference src/sycl/SiluAndMulClamp.cpp:61-74 (both halves clamped in bf16, then fp32 silu * up).
// The same stream as act_and_mul_kernel: one 16-byte access per lane and half.
__device__ __forceinline__ float through_bf16(float v) { return (float)(bf16)v; }

template <typename T, int VEC>
__global__ __launch_bounds__(256) void silu_mul_clamp_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                             int64_t total_vecs, int dvec, float limit) {
  const float lim = through_bf16(limit);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_vecs; idx += stride) {
    const int64_t tok = idx / dvec;
    const int j = (int)(idx - tok * dvec);
    const T* row = x + tok * (int64_t)dvec * 2 * VEC;
    Vec<T, VEC> a = load_vec<T, VEC>(row + (int64_t)j * VEC);
    Vec<T, VEC> b = load_vec<T, VEC>(row + ((int64_t)dvec + j) * VEC);
    Vec<T, VEC> y;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      // (the reference rounds the clamped values to bf16 once more: min / max of two bf16 values is one of them; and a
      // bf16 input is its own rounding)
      const float ga = sizeof(T) == 2 && __is_same(T, bf16) ? (float)a[e] : through_bf16((float)a[e]);
      const float ub = sizeof(T) == 2 && __is_same(T, bf16) ? (float)b[e] : through_bf16((float)b[e]);
      const float g = fminf(ga, lim);
      const float u = fmaxf(-lim, fminf(ub, lim));
      y[e] = (T)(g * (1.0f / (1.0f + expf(-g))) * u);
    }
    store_vec<T, VEC>(out + idx * VEC, y);
  }
}

template <typename T, int VEC>
static int launch_clamp(hipStream_t st, T* out, const T* x, int64_t tokens, int64_t d, float limit) {
  const int64_t dvec = d / VEC, total = tokens * dvec, want = cdiv(total, 256);
  const unsigned blocks = (unsigned)(want < 256 * 16 ? want : 256 * 16);
  silu_mul_clamp_kernel<T, VEC><<<blocks, 256, 0, st>>>(out, x, total, (int)dvec, limit);
  return check_launch("silu_and_mul_clamp");
}

// ---- swiglu_gpt_oss_sigmoid_alpha: reference src/sycl/SwigluAlphaLimit.cpp:16-98. x is a flat stream of (gate, up) pairs;
// a lane reads PAIRS of them with one access (16 bytes of 16-bit pairs, 16 or 32 bytes of fp32 pairs) and writes PAIRS
// outputs with one store. PAIRS = 1 is the unaligned / odd-count form.
__device__ __forceinline__ float swiglu_pair(float gate, float up, float alpha, float limit) {
  gate = fminf(gate, limit);
  up = fmaxf(-limit, fminf(up, limit));
  const float sig = 1.0f / (1.0f + expf(-(gate * alpha)));
  return gate * sig * (up + 1.0f);
}

template <typename T, int PAIRS>
__global__ __launch_bounds__(256) void swiglu_alpha_limit_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                                 int64_t total_groups, float alpha, float limit) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_groups; idx += stride) {
    Vec<T, PAIRS> y;
    if constexpr (PAIRS * 2 * sizeof(T) <= 16) {
      const Vec<T, 2 * PAIRS> v = load_vec<T, 2 * PAIRS>(x + idx * (2 * PAIRS));
#pragma unroll
      for (int e = 0; e < PAIRS; ++e) y[e] = (T)swiglu_pair((float)v[2 * e], (float)v[2 * e + 1], alpha, limit);
    } else {  // (fp32, 4 pairs: two 16-byte reads)
      const Vec<T, PAIRS> lo = load_vec<T, PAIRS>(x + idx * (2 * PAIRS));
      const Vec<T, PAIRS> hi = load_vec<T, PAIRS>(x + idx * (2 * PAIRS) + PAIRS);
#pragma unroll
      for (int e = 0; e < PAIRS / 2; ++e) {
        y[e] = (T)swiglu_pair((float)lo[2 * e], (float)lo[2 * e + 1], alpha, limit);
        y[PAIRS / 2 + e] = (T)swiglu_pair((float)hi[2 * e], (float)hi[2 * e + 1], alpha, limit);
      }
    }
    store_vec<T, PAIRS>(out + idx * PAIRS, y);
  }
}

template <typename T, int PAIRS>
static int launch_swiglu(hipStream_t st, T* out, const T* x, int64_t pairs, float alpha, float limit) {
  const int64_t total = pairs / PAIRS, want = cdiv(total, 256);
  const unsigned blocks = (unsigned)(want < 256 * 16 ? want : 256 * 16);
  swiglu_alpha_limit_kernel<T, PAIRS><<<blocks, 256, 0, st>>>(out, x, total, alpha, limit);
  return check_launch("swiglu_gpt_oss_sigmoid_alpha");
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_silu_and_mul_clamp(sglk_stream_t stream, void* out, const void* x, int64_t tokens, int64_t d,
                                       int dtype, float limit) {
  using namespace sglk;
  SGLK_REQUIRE(tokens >= 0 && d > 0 && d < (1ll << 30), "silu_and_mul_clamp: bad shape tokens=%lld d=%lld",
               (long long)tokens, (long long)d);
  SGLK_REQUIRE(dtype == SGLK_BF16 || dtype == SGLK_F16, "silu_and_mul_clamp: input must be Half or BFloat16");
  SGLK_REQUIRE(limit > 0.f, "silu_and_mul_clamp: swiglu_limit must be > 0");
  if (tokens == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  SGLK_DISPATCH_HALF(dtype, T, {
    int v = 8;
    while (v > 1 && (d % v || (uintptr_t)x % (v * sizeof(T)) || (uintptr_t)out % (v * sizeof(T)))) v >>= 1;
    if (v == 8) return launch_clamp<T, 8>(st, (T*)out, (const T*)x, tokens, d, limit);
    if (v >= 2) return launch_clamp<T, 2>(st, (T*)out, (const T*)x, tokens, d, limit);
    return launch_clamp<T, 1>(st, (T*)out, (const T*)x, tokens, d, limit);
  });
  return SGLK_OK;
}

extern "C" int sglk_swiglu_alpha_limit(sglk_stream_t stream, void* out, const void* x, int64_t rows, int64_t hidden,
                                       int dtype, float alpha, float limit) {
  using namespace sglk;
  SGLK_REQUIRE(rows >= 0 && hidden > 0 && hidden < (1ll << 30), "swiglu_gpt_oss_sigmoid_alpha: bad shape rows=%lld hidden=%lld",
               (long long)rows, (long long)hidden);
  SGLK_REQUIRE(limit > 0.f, "swiglu_gpt_oss_sigmoid_alpha: gemm1_limit must be positive");
  if (rows == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t pairs = rows * hidden;
  SGLK_DISPATCH_FLOAT(dtype, T, {
    const bool vec = pairs % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % (4 * sizeof(T)) == 0;
    if (vec) return launch_swiglu<T, 4>(st, (T*)out, (const T*)x, pairs, alpha, limit);
    return launch_swiglu<T, 1>(st, (T*)out, (const T*)x, pairs, alpha, limit);
  });
  return SGLK_OK;
}

extern "C" int sglk_act_and_mul(sglk_stream_t stream, void* out, const void* x, int64_t tokens,
                                int64_t d, int dtype, int act) {
  using namespace sglk;
  SGLK_REQUIRE(tokens >= 0 && d > 0 && d < (1ll << 30), "act_and_mul: bad shape tokens=%lld d=%lld",
               (long long)tokens, (long long)d);
  if (tokens == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  SGLK_DISPATCH_FLOAT(dtype, T, {
    constexpr int kMax = 16 / sizeof(T);
    int v = kMax;
    while (v > 1 && (d % v || (uintptr_t)x % (v * sizeof(T)) || (uintptr_t)out % (v * sizeof(T)))) v >>= 1;
    if (v == kMax) return launch<T, kMax>(st, (T*)out, (const T*)x, tokens, d, act);
    if (v >= 2) return launch<T, 2>(st, (T*)out, (const T*)x, tokens, d, act);
    return launch<T, 1>(st, (T*)out, (const T*)x, tokens, d, act);
  });
  return SGLK_OK;
}
