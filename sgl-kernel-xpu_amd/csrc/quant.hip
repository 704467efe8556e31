// sgl_per_token_group_quant_8bit (fp8 e4m3fn / int8) for gfx950.
//
// Replaces reference src/sycl/per_token_group_quant_8bit.cpp:48-220 (kernel),
// :222-386 (host). Arithmetic kept, step for step (all fp32, IEEE div):
//   amax = max(eps, |x_i|)                       (:97, :139-141)
//   y_s  = amax / qmax                           (:157)
//   ue8m0: e = ceil(log2(max(y_s,1e-10))); y_s = 2^e; byte = e + 127   (:161-165)
//   q_i  = cast(min(max(x_i * (1/y_s), qmin), qmax))                  (:171, :185)
//   fp8 cast: round-to-nearest-even; int8 cast: truncation           (:191-196)
// The ue8m0 exponent is taken from the float's bit pattern (exact ceil(log2)),
// where the reference calls log2/ceil; they agree wherever log2 is exact.
// Scale placement reproduces :106-125 (row-major, column-major, packed ue8m0).
//
// Design: HBM stream, 3 B/element. Each lane owns 16 consecutive elements
// (2x or 4x 16-byte loads -> one 16-byte store); a group of G elements is G/16
// adjacent lanes, reduced with xor-shuffles that never leave the group.
#include "common.h"

namespace sglk {
namespace {

constexpr int kEPL = 16;  // elements per lane

template <typename T>
__device__ __forceinline__ void load16(const T* p, float (&f)[kEPL]) {
  constexpr int V = 16 / sizeof(T);
#pragma unroll
  for (int i = 0; i < kEPL / V; ++i) {
    Vec<T, V> v = load_vec<T, V>(p + i * V);
#pragma unroll
    for (int j = 0; j < V; ++j) f[i * V + j] = (float)v[j];
  }
}

template <typename T, int GROUP, bool FP8, int SCALE_KIND>
__global__ __launch_bounds__(256) void group_quant_kernel(const T* __restrict__ x,
                                                          uint8_t* __restrict__ q,
                                                          void* __restrict__ scales,
                                                          int64_t num_groups, int groups_per_row,
                                                          float eps, float qmin, float qmax,
                                                          int64_t s_stride_row, int64_t s_stride_col) {
  constexpr int LPG = GROUP / kEPL;  // lanes per group (2..32)
  const int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gid = gl / LPG;
  const int lig = (int)(gl % LPG);
  if (gid >= num_groups) return;  // whole groups leave together: shuffles below stay inside a group

  float f[kEPL];
  load16<T>(x + gid * GROUP + lig * kEPL, f);

  float amax = eps;
#pragma unroll
  for (int i = 0; i < kEPL; ++i) amax = fmaxf(amax, fabsf(f[i]));
  amax = group_max<LPG>(amax);

  float y_s = amax / qmax;
  uint32_t ue8 = 0;
  if constexpr (SCALE_KIND != 0) {
    const float c = fmaxf(y_s, 1e-10f);
    const uint32_t bits = __float_as_uint(c);
    const int e = (int)((bits >> 23) & 0xff) - 127 + ((bits & 0x7fffffu) != 0);
    ue8 = (uint32_t)(e + 127);
    y_s = __uint_as_float(ue8 << 23);
  }

  if (lig == 0) {
    const int64_t row = gid / groups_per_row;
    const int64_t g = gid - row * groups_per_row;
    if constexpr (SCALE_KIND == 0) {
      ((float*)scales)[row * s_stride_row + g * s_stride_col] = y_s;
    } else if constexpr (SCALE_KIND == 1) {
      ((uint8_t*)scales)[gid] = (uint8_t)ue8;
    } else {
      ((uint8_t*)scales)[((g >> 2) * s_stride_col + row) * 4 + (g & 3)] = (uint8_t)ue8;
    }
  }

  const float inv = 1.0f / y_s;
  Vec<uint32_t, 4> o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(f[i * 4 + j] * inv, qmin), qmax);
    if constexpr (FP8) {
      o[i] = pack4_e4m3(v[0], v[1], v[2], v[3]);
    } else {
      o[i] = ((uint32_t)(uint8_t)(int8_t)(int)v[0]) | ((uint32_t)(uint8_t)(int8_t)(int)v[1] << 8) |
             ((uint32_t)(uint8_t)(int8_t)(int)v[2] << 16) | ((uint32_t)(uint8_t)(int8_t)(int)v[3] << 24);
    }
  }
  store_vec<uint32_t, 4>((uint32_t*)(q + gid * GROUP + lig * kEPL), o);
}

template <typename T, int GROUP, bool FP8>
static int launch_kind(hipStream_t st, const T* x, uint8_t* q, void* s, int64_t num_groups,
                       int groups_per_row, float eps, float qmin, float qmax, int scale_kind,
                       int64_t ssr, int64_t ssc) {
  constexpr int LPG = GROUP / kEPL;
  const int64_t lanes = num_groups * LPG;
  const unsigned blocks = (unsigned)cdiv(lanes, 256);
  switch (scale_kind) {
    case 0:
      group_quant_kernel<T, GROUP, FP8, 0><<<blocks, 256, 0, st>>>(x, q, s, num_groups, groups_per_row,
                                                                  eps, qmin, qmax, ssr, ssc);
      break;
    case 1:
      group_quant_kernel<T, GROUP, FP8, 1><<<blocks, 256, 0, st>>>(x, q, s, num_groups, groups_per_row,
                                                                  eps, qmin, qmax, ssr, ssc);
      break;
    case 2:
      group_quant_kernel<T, GROUP, FP8, 2><<<blocks, 256, 0, st>>>(x, q, s, num_groups, groups_per_row,
                                                                  eps, qmin, qmax, ssr, ssc);
      break;
    default:
      return fail(SGLK_EINVAL, "per_token_group_quant_8bit: unknown scale_kind %d", scale_kind);
  }
  return check_launch("per_token_group_quant_8bit");
}

template <typename T, bool FP8>
static int launch_group(hipStream_t st, const T* x, uint8_t* q, void* s, int64_t num_groups,
                        int groups_per_row, int group, float eps, float qmin, float qmax, int kind,
                        int64_t ssr, int64_t ssc) {
  switch (group) {
#define SGLK_CASE(G) \
  case G:            \
    return launch_kind<T, G, FP8>(st, x, q, s, num_groups, groups_per_row, eps, qmin, qmax, kind, ssr, ssc);
    SGLK_CASE(32)
    SGLK_CASE(64)
    SGLK_CASE(128)
    SGLK_CASE(256)
    SGLK_CASE(512)
#undef SGLK_CASE
    default:
      return fail(SGLK_EUNSUPPORTED,
                  "per_token_group_quant_8bit: unsupported group_size %d (supported: 32, 64, 128, 256, 512)",
                  group);
  }
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_per_token_group_quant_8bit(sglk_stream_t stream, const void* x, void* q,
                                               void* scales, int64_t rows, int64_t k, int group_size,
                                               float eps, float qmin, float qmax, int in_dtype,
                                               int out_dtype, int scale_kind, int64_t s_stride_row,
                                               int64_t s_stride_col) {
  using namespace sglk;
  SGLK_REQUIRE(rows >= 0 && k > 0, "per_token_group_quant_8bit: bad shape rows=%lld k=%lld",
               (long long)rows, (long long)k);
  SGLK_REQUIRE(group_size > 0 && k % group_size == 0,
               "per_token_group_quant_8bit: hidden size %lld not divisible by group_size %d", (long long)k,
               group_size);
  SGLK_REQUIRE(out_dtype == SGLK_FP8_E4M3 || out_dtype == SGLK_INT8,
               "per_token_group_quant_8bit: output_q dtype must be Int8 or Float8_e4m3fn");
  SGLK_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 16) == 0,
               "per_token_group_quant_8bit: input/output pointers must be 16-byte aligned");
  if (rows == 0) return SGLK_OK;
  const int64_t num_groups = rows * (k / group_size);
  const int gpr = (int)(k / group_size);
  hipStream_t st = (hipStream_t)stream;
  SGLK_DISPATCH_FLOAT(in_dtype, T, {
    if (out_dtype == SGLK_FP8_E4M3)
      return launch_group<T, true>(st, (const T*)x, (uint8_t*)q, scales, num_groups, gpr, group_size, eps,
                                   qmin, qmax, scale_kind, s_stride_row, s_stride_col);
    return launch_group<T, false>(st, (const T*)x, (uint8_t*)q, scales, num_groups, gpr, group_size, eps,
                                  qmin, qmax, scale_kind, s_stride_row, s_stride_col);
  });
  return SGLK_OK;
}
