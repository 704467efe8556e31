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

}  // namespace
}  // namespace sglk

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
