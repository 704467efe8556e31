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
#include <math.h>

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

// ---- v2: + fused silu-and-mul, + expert-masked layout (reference
// src/sycl/per_token_group_quant_8bit_v2.cpp:160-332 compute, :714-842 host).
//   fuse_silu_and_mul: x is [.., 2H]; the value quantised is T(T(silu(x1)) * x2) with
//                      silu(v) = h (1 + tanh(h)), h = v/2                         (:113-117, :257-259)
//   masked_m:          x is [E, T_pad, *]; only rows < masked_m[e] of expert e are processed.
// Scale (row, group) of expert e goes to output_s[e, row, group] through the tensor's strides (float), or as a
// UE8M0 byte: row-major [.., G] or packed 4-per-int32 column-major (bytes past the last group of a partly
// filled pack are zeroed, :208-222). NOTE: the reference's column-major branch places scale (row, group) at the
// coordinates divmod(row + group * T_pad, G) (:173-193); this build writes the logical element [row, group],
// which is what the consumers of a column-major [T, G] scale tensor index.
template <typename T, int GROUP, bool FP8, int SCALE_KIND, bool FUSE>
__global__ __launch_bounds__(256) void group_quant_v2_kernel(const T* __restrict__ x, uint8_t* __restrict__ q,
                                                             void* __restrict__ scales,
                                                             const int32_t* __restrict__ masked_m, int64_t num_groups,
                                                             int groups_per_row, int rows_per_expert, float eps,
                                                             float qmin, float qmax, int64_t s_stride_e,
                                                             int64_t s_stride_row, int64_t s_stride_col) {
  constexpr int LPG = GROUP / kEPL;
  const int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gid = gl / LPG;
  const int lig = (int)(gl % LPG);
  if (gid >= num_groups) return;
  const int64_t row_all = gid / groups_per_row;       // (expert, row) flattened
  const int g = (int)(gid - row_all * groups_per_row);
  const int e = (int)(row_all / rows_per_expert);
  const int row = (int)(row_all - (int64_t)e * rows_per_expert);
  if (masked_m != nullptr && row >= masked_m[e]) return;
  const int hidden = groups_per_row * GROUP;

  float f[kEPL];
  if constexpr (FUSE) {
    const T* base = x + row_all * (2 * (int64_t)hidden) + g * GROUP + lig * kEPL;
    float a[kEPL], b[kEPL];
    load16<T>(base, a);
    load16<T>(base + hidden, b);
#pragma unroll
    for (int i = 0; i < kEPL; ++i) {
      const float h = 0.5f * a[i];
      const T sv = (T)(h * (1.0f + tanhf(h)));
      f[i] = (float)(T)((float)sv * b[i]);
    }
  } else {
    load16<T>(x + row_all * (int64_t)hidden + g * GROUP + lig * kEPL, f);
  }

  float amax = eps;
#pragma unroll
  for (int i = 0; i < kEPL; ++i) amax = fmaxf(amax, fabsf(f[i]));
  amax = group_max<LPG>(amax);

  float y_s = amax / qmax;
  uint32_t ue8 = 0;
  if constexpr (SCALE_KIND != 0) {
    const float c = fmaxf(y_s, 1e-10f);
    const uint32_t bits = __float_as_uint(c);
    const int ex = (int)((bits >> 23) & 0xff) - 127 + ((bits & 0x7fffffu) != 0);
    ue8 = (uint32_t)(ex + 127);
    y_s = __uint_as_float(ue8 << 23);
  }
  if (lig == 0) {
    if constexpr (SCALE_KIND == 0) {
      ((float*)scales)[e * s_stride_e + row * s_stride_row + g * s_stride_col] = y_s;
    } else if constexpr (SCALE_KIND == 1) {
      ((uint8_t*)scales)[row_all * groups_per_row + g] = (uint8_t)ue8;
    } else {
      uint8_t* p = (uint8_t*)scales + (e * s_stride_e + (g >> 2) * s_stride_col + row) * 4 + (g & 3);
      *p = (uint8_t)ue8;
      if (g == groups_per_row - 1)
        for (int i = (g & 3) + 1; i < 4; ++i) p[i - (g & 3)] = 0;
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

struct QuantV2Args {
  const void* x; uint8_t* q; void* s; const int32_t* masked_m;
  int64_t num_groups; int groups_per_row, rows_per_expert; float eps, qmin, qmax;
  int64_t sse, ssr, ssc;
};

template <typename T, int GROUP, bool FP8, bool FUSE>
static int launch_v2_kind(hipStream_t st, const QuantV2Args& a, int kind) {
  constexpr int LPG = GROUP / kEPL;
  const unsigned blocks = (unsigned)cdiv(a.num_groups * LPG, 256);
#define SGLK_GO(KIND)                                                                                              \
  group_quant_v2_kernel<T, GROUP, FP8, KIND, FUSE><<<blocks, 256, 0, st>>>(                                         \
      (const T*)a.x, a.q, a.s, a.masked_m, a.num_groups, a.groups_per_row, a.rows_per_expert, a.eps, a.qmin, a.qmax, \
      a.sse, a.ssr, a.ssc)
  switch (kind) {
    case 0: SGLK_GO(0); break;
    case 1: SGLK_GO(1); break;
    case 2: SGLK_GO(2); break;
    default: return fail(SGLK_EINVAL, "per_token_group_quant_8bit_v2: unknown scale_kind %d", kind);
  }
#undef SGLK_GO
  return check_launch("per_token_group_quant_8bit_v2");
}

template <typename T, bool FP8, bool FUSE>
static int launch_v2_group(hipStream_t st, const QuantV2Args& a, int group, int kind) {
  switch (group) {
    case 16: return launch_v2_kind<T, 16, FP8, FUSE>(st, a, kind);
    case 32: return launch_v2_kind<T, 32, FP8, FUSE>(st, a, kind);
    case 64: return launch_v2_kind<T, 64, FP8, FUSE>(st, a, kind);
    case 128: return launch_v2_kind<T, 128, FP8, FUSE>(st, a, kind);
    default: return fail(SGLK_EUNSUPPORTED, "Unsupported group_size");
  }
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

extern "C" int sglk_per_token_group_quant_8bit_v2(sglk_stream_t stream, const void* x, void* q, void* scales,
                                                  const int32_t* masked_m, int64_t num_experts,
                                                  int64_t rows_per_expert, int64_t hidden, int group_size, float eps,
                                                  float qmin, float qmax, int in_dtype, int out_dtype, int scale_kind,
                                                  int64_t s_stride_expert, int64_t s_stride_row, int64_t s_stride_col,
                                                  int fuse_silu_and_mul) {
  using namespace sglk;
  SGLK_REQUIRE(num_experts > 0 && rows_per_expert >= 0 && hidden > 0, "per_token_group_quant_8bit_v2: bad shape");
  SGLK_REQUIRE(group_size > 0 && hidden % group_size == 0, "per_token_group_quant_8bit_v2: hidden size %lld not divisible by group_size %d",
               (long long)hidden, group_size);
  SGLK_REQUIRE(in_dtype == SGLK_F16 || in_dtype == SGLK_BF16, "per_token_group_quant_8bit_v2: input must be Half or BFloat16");
  SGLK_REQUIRE(out_dtype == SGLK_FP8_E4M3 || out_dtype == SGLK_INT8, "per_token_group_quant_8bit_v2: output_q dtype must be Int8 or Float8_e4m3fn");
  SGLK_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 16) == 0, "per_token_group_quant_8bit_v2: pointers must be 16-byte aligned");
  if (rows_per_expert == 0) return SGLK_OK;
  QuantV2Args a{x, (uint8_t*)q, scales, masked_m, num_experts * rows_per_expert * (hidden / group_size),
                (int)(hidden / group_size), (int)rows_per_expert, eps, qmin, qmax, s_stride_expert, s_stride_row,
                s_stride_col};
  hipStream_t st = (hipStream_t)stream;
  const bool fp8 = out_dtype == SGLK_FP8_E4M3;
  SGLK_DISPATCH_HALF(in_dtype, T, {
    if (fuse_silu_and_mul) {
      if (fp8) return launch_v2_group<T, true, true>(st, a, group_size, scale_kind);
      return launch_v2_group<T, false, true>(st, a, group_size, scale_kind);
    }
    if (fp8) return launch_v2_group<T, true, false>(st, a, group_size, scale_kind);
    return launch_v2_group<T, false, false>(st, a, group_size, scale_kind);
  });
  return SGLK_OK;
}
