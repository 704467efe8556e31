// RMSNorm family for gfx950: rmsnorm, gemma_rmsnorm, fused_add_rmsnorm,
// gemma_fused_add_rmsnorm.
//
// Replaces reference src/sycl/RMSNorm.cpp:793-905 (+ Norm.h). Behaviour kept:
//   rmsnorm:        y = T((w * rstd) * x)              (RMSNorm.cpp:105-136)
//   gemma_rmsnorm:  y = T((x * rstd) * (1 + w))        (RMSNorm.cpp:436-446)
//   fused_add:      r = T(x + res) stored to BOTH x-row and residual, variance
//                   over the rounded r                 (RMSNorm.cpp:160-181)
//   rstd = rsqrt(max(sum,0)/n + eps), fp32 accumulation.
// Design (HBM-bound, one pass): every row is read once into registers (up to
// 8 x 16-byte vectors per lane), reduced with wave shuffles (+ one LDS hop for
// rows wider than a wave can hold), scaled and written once. Unlike the
// reference there is no rstd temp in global memory and no second read of x.
//   n <= 64*8*VEC elements : one wave per row, 4 rows per 256-thread block
//   n <= 256*8*VEC          : one 256-thread block per row
//   n <= 1024*8*VEC         : one 1024-thread block per row
//   beyond                  : 1024-thread block, second pass re-reads x (L2)
#include <initializer_list>

#include "common.h"

namespace sglk {
namespace {

constexpr int kMaxCache = 8;

__device__ __forceinline__ int64_t row_offset(int64_t row, const sglk_row_strides& s) {
  return (row / s.inner_size) * s.outer_stride + (row % s.inner_size) * s.inner_stride;
}

template <typename T, typename W, int VEC, int TPR, bool GEMMA, bool ADD>
__global__ __launch_bounds__((TPR < 256 ? 256 : TPR)) void rmsnorm_kernel(
    T* out, const T* x, T* residual, const W* __restrict__ w, int64_t rows, int n,
    sglk_row_strides xs, sglk_row_strides os, float eps) {
  constexpr int BLOCK = TPR < 256 ? 256 : TPR;
  constexpr int RPB = BLOCK / TPR;
  constexpr int NW = TPR / 64;
  __shared__ float red[NW > 1 ? NW : 1];

  const int t = threadIdx.x % TPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;
  if (row >= rows) return;  // RPB > 1 only when TPR == 64: no block barrier is used then

  const int nvec = n / VEC;
  const T* xr = x + row_offset(row, xs);
  T* yr = out + row_offset(row, os);
  T* rr = ADD ? residual + row_offset(row, xs) : nullptr;
  const bool cached = nvec <= TPR * kMaxCache;

  Vec<T, VEC> cache[kMaxCache];
  float ss = 0.f;
  if (cached) {
#pragma unroll
    for (int c = 0; c < kMaxCache; ++c) {
      const int i = t + c * TPR;
      if (i < nvec) {
        cache[c] = load_vec<T, VEC>(xr + (int64_t)i * VEC);
        if constexpr (ADD) {
          Vec<T, VEC> r = load_vec<T, VEC>(rr + (int64_t)i * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) cache[c][j] = (T)((float)cache[c][j] + (float)r[j]);
          store_vec<T, VEC>(rr + (int64_t)i * VEC, cache[c]);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float f = (float)cache[c][j];
          ss += f * f;
        }
      }
    }
  } else {
    for (int i = t; i < nvec; i += TPR) {
      Vec<T, VEC> v = load_vec<T, VEC>(xr + (int64_t)i * VEC);
      if constexpr (ADD) {
        Vec<T, VEC> r = load_vec<T, VEC>(rr + (int64_t)i * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = (T)((float)v[j] + (float)r[j]);
        store_vec<T, VEC>(rr + (int64_t)i * VEC, v);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float f = (float)v[j];
        ss += f * f;
      }
    }
  }

  ss = block_sum<NW>(ss, red);
  const float rstd = rsqrtf(fmaxf(ss, 0.f) / (float)n + eps);

  auto emit = [&](int i, const Vec<T, VEC>& xv) {
    Vec<W, VEC> wv = load_vec<W, VEC>(w + (int64_t)i * VEC);
    Vec<T, VEC> y;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if constexpr (GEMMA) {
        y[j] = (T)(((float)xv[j] * rstd) * (1.0f + (float)wv[j]));
      } else {
        y[j] = (T)(((float)wv[j] * rstd) * (float)xv[j]);
      }
    }
    store_vec<T, VEC>(yr + (int64_t)i * VEC, y);
  };

  if (cached) {
#pragma unroll
    for (int c = 0; c < kMaxCache; ++c) {
      const int i = t + c * TPR;
      if (i < nvec) emit(i, cache[c]);
    }
  } else {
    // second pass: for ADD the summed row was just written to residual by this thread
    const T* src = ADD ? rr : xr;
    for (int i = t; i < nvec; i += TPR) emit(i, load_vec<T, VEC>(src + (int64_t)i * VEC));
  }
}

static int vec_for(int64_t max_vec, int64_t n, int64_t esz, std::initializer_list<uintptr_t> addrs,
                   std::initializer_list<int64_t> strides_elems) {
  int64_t v = max_vec;
  auto ok = [&](int64_t v) {
    if (n % v) return false;
    for (auto a : addrs)
      if (a % (uintptr_t)(v * esz)) return false;
    for (auto s : strides_elems)
      if (s % v) return false;
    return true;
  };
  while (v > 1 && !ok(v)) v >>= 1;
  return (int)v;
}

template <typename T, typename W, int VEC, bool GEMMA, bool ADD>
static int launch_vec(hipStream_t st, T* out, const T* x, T* res, const W* w, int64_t rows, int64_t n,
                      sglk_row_strides xs, sglk_row_strides os, float eps) {
  const int64_t nvec = n / VEC;
  // (one wave per row only when there are rows enough to fill the chip with waves: a size sweep - round 5, late - found decode-sized
  //  calls at hidden 4096 slower than at 4104, 4.7 - 5.8 us against 3.5 - 4.0 for 1 .. 256 rows and 6.6 against 5.4 at 1024; at 4096
  //  rows the wave form wins, 12.9 against 16.2 us)
  if (nvec <= 64 * kMaxCache && rows >= 2048) {
    const int64_t blocks = cdiv(rows, 4);
    rmsnorm_kernel<T, W, VEC, 64, GEMMA, ADD><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
        out, x, res, w, rows, (int)n, xs, os, eps);
  } else if (nvec <= 256 * kMaxCache) {
    rmsnorm_kernel<T, W, VEC, 256, GEMMA, ADD><<<dim3((unsigned)rows), dim3(256), 0, st>>>(
        out, x, res, w, rows, (int)n, xs, os, eps);
  } else {
    rmsnorm_kernel<T, W, VEC, 1024, GEMMA, ADD><<<dim3((unsigned)rows), dim3(1024), 0, st>>>(
        out, x, res, w, rows, (int)n, xs, os, eps);
  }
  return check_launch("rmsnorm");
}

template <typename T, typename W, bool GEMMA, bool ADD>
static int launch(hipStream_t st, void* out, const void* x, void* res, const void* w, int64_t rows,
                  int64_t n, sglk_row_strides xs, sglk_row_strides os, float eps) {
  constexpr int64_t kMaxVec = 16 / sizeof(T);
  // weight vectors are VEC elements of W: their alignment is checked against sizeof(W)
  int v = vec_for(kMaxVec, n, sizeof(T), {(uintptr_t)x, (uintptr_t)out, (uintptr_t)res},
                  {xs.outer_stride, xs.inner_stride, os.outer_stride, os.inner_stride});
  while (v > 1 && ((uintptr_t)w % (uintptr_t)(v * sizeof(W)))) v >>= 1;
  T* o = (T*)out;
  const T* xi = (const T*)x;
  T* r = (T*)res;
  const W* wi = (const W*)w;
  switch (v) {
    case 8:
      if constexpr (kMaxVec >= 8) return launch_vec<T, W, 8, GEMMA, ADD>(st, o, xi, r, wi, rows, n, xs, os, eps);
    case 4:
      return launch_vec<T, W, 4, GEMMA, ADD>(st, o, xi, r, wi, rows, n, xs, os, eps);
    case 2:
      return launch_vec<T, W, 2, GEMMA, ADD>(st, o, xi, r, wi, rows, n, xs, os, eps);
    default:
      return launch_vec<T, W, 1, GEMMA, ADD>(st, o, xi, r, wi, rows, n, xs, os, eps);
  }
}

template <bool ADD>
static int dispatch(hipStream_t st, void* out, const void* x, void* res, const void* w, int64_t rows,
                    int64_t n, sglk_row_strides xs, sglk_row_strides os, float eps, int dtype,
                    int wdtype, int gemma) {
  SGLK_REQUIRE(rows >= 0 && n > 0, "rmsnorm: bad shape rows=%lld n=%lld", (long long)rows, (long long)n);
  SGLK_REQUIRE(n < (1ll << 31), "rmsnorm: hidden size too large");
  SGLK_REQUIRE(xs.inner_size > 0 && os.inner_size > 0, "rmsnorm: inner_size must be positive");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_FLOAT(dtype, T, {
    SGLK_DISPATCH_FLOAT(wdtype, W, {
      if (gemma) return launch<T, W, true, ADD>(st, out, x, res, w, rows, n, xs, os, eps);
      return launch<T, W, false, ADD>(st, out, x, res, w, rows, n, xs, os, eps);
    });
  });
  return SGLK_OK;
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_rmsnorm(sglk_stream_t stream, void* out, const void* x, const void* weight,
                            int64_t rows, int64_t n, sglk_row_strides x_strides,
                            sglk_row_strides out_strides, float eps, int dtype, int weight_dtype,
                            int gemma) {
  return sglk::dispatch<false>((hipStream_t)stream, out, x, nullptr, weight, rows, n, x_strides,
                               out_strides, eps, dtype, weight_dtype, gemma);
}

extern "C" int sglk_fused_add_rmsnorm(sglk_stream_t stream, void* x, void* residual,
                                      const void* weight, int64_t rows, int64_t n, float eps,
                                      int dtype, int weight_dtype, int gemma) {
  sglk_row_strides s{n, 1, 0};
  return sglk::dispatch<true>((hipStream_t)stream, x, x, residual, weight, rows, n, s, s, eps, dtype,
                              weight_dtype, gemma);
}
