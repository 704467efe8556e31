// Fused per-head RMSNorm + rotary embedding of q and k, in place.
//
// Replaces reference src/sycl/FusedQKNormRope.cpp:
//   fused_qk_norm_rope        (:507-615, kernel FusedQKNormRopeKernel :268-398; schema torch_extension_sycl.cc:416-420):
//       packed qkv [tokens, (Hq + Hk + Hv) * D], rotary angles computed on the fly from `base` with optional YaRN
//       frequency blending (computeFreqYarn :42-67) and an attention factor on the rotated part
//   fused_inplace_qknorm_rope (:1723-1861, kernel FusedQKNormRopeCacheKernel :617-737; schema :421-424):
//       separate q [tokens, Hq, D] / k [tokens, Hk, D] (token and head strides free), angles from a precomputed fp32
//       cos_sin_cache [max_pos, rope_dim] (cos first half, sin second half)
// Arithmetic of both (fp32 throughout, one rounding to T at the end):
//   y[d] = x[d] * (rsqrt(mean_d x^2 + eps) * w[d])
//   neox:        d <  rope/2: y[d] c - y[d + rope/2] s ;  rope/2 <= d < rope: y[d] c + y[d - rope/2] s ;  c, s at d mod rope/2
//   interleaved: (y[2j], y[2j+1]) -> (y[2j] c - y[2j+1] s, y[2j] s + y[2j+1] c) ;  c, s at j
//   d >= rope: y[d] unchanged. V heads are never touched.
//
// Kernel: D / 8 lanes per (token, head) row, 8 elements (16 bytes of a 16-bit type) per lane, 64 / (D / 8) rows per
// wave; the sum of squares is an xor-shuffle reduction inside the lane group. The neox partner elements are not
// shuffled between lanes but re-read (they sit in the cache lines the wave has just loaded) and normalised with the
// same factor: no LDS, no cross-lane layout constraint on rope_dim; all loads of a row precede its stores in the one
// wave that owns it. One HBM pass: 2 x tokens x (Hq + Hk) x D x sizeof(T) bytes.
#include "common.h"

namespace sglk {
namespace {

struct RopeParams {
  const float* cos_sin_cache;  // cache mode
  const void* positions;
  int pos_is_i64;
  int rope_dim;
  int cs_vec;  // the cached angles of a 16-byte chunk can be read as 16-byte loads (set by launch_all)
  float eps;
  // analytic mode
  float log2_base, factor, low, high, attention_factor;
};

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&x)[8]) {
  if constexpr (sizeof(T) == 2) {
    const Vec<T, 8> v = load_vec<T, 8>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)v[i];
  } else {
    const Vec<T, 4> a = load_vec<T, 4>(p), b = load_vec<T, 4>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = (float)a[i]; x[4 + i] = (float)b[i]; }
  }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&x)[8]) {
  if constexpr (sizeof(T) == 2) {
    Vec<T, 8> v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)x[i];
    store_vec<T, 8>(p, v);
  } else {
    Vec<T, 4> a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = (T)x[i]; b[i] = (T)x[4 + i]; }
    store_vec<T, 4>(p, a);
    store_vec<T, 4>(p + 4, b);
  }
}

// reference computeFreqYarn (FusedQKNormRope.cpp:42-67)
__device__ __forceinline__ float yarn_freq(float log2_base, int rotary_dim, int half_dim, float factor, float low, float high) {
  const float exponent = -2.0f * (float)half_dim / (float)rotary_dim;
  float freq = exp2f(exponent * log2_base);
  if (factor != 1.0f) {
    const float extrapolation = freq, interpolation = freq / factor;
    float high_adj = high;
    if (fabsf(low - high_adj) <= 1e-6f) high_adj += 0.001f;
    const float linear = (2.0f * (float)half_dim - low) / (high_adj - low);
    const float ramp = fminf(fmaxf(linear, 0.0f), 1.0f);
    freq = interpolation * (1.0f - ramp) + extrapolation * ramp;
  }
  return freq;
}

template <typename T, int D, bool NEOX, bool ANALYTIC>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(T* __restrict__ q, T* __restrict__ k, const T* __restrict__ qw,
                                                          const T* __restrict__ kw, RopeParams p, int64_t tokens, int Hq,
                                                          int Hk, int64_t q_ts, int64_t q_hs, int64_t k_ts, int64_t k_hs) {
  constexpr int G = D / 8;        // lanes per row
  constexpr int RW = 256 / G;     // rows per workgroup
  const int l = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * RW + threadIdx.x / G;
  const int heads = Hq + Hk;
  const bool live = row < tokens * heads;
  const int64_t rr = live ? row : 0;
  // (a 64-bit division is ~100 vector instructions per thread - more than the rest of the kernel; the row count fits 32 bits in
  //  every call a server makes)
  const int64_t tok = (tokens * heads < (1ll << 32)) ? (int64_t)((uint32_t)rr / (uint32_t)heads) : rr / heads;
  const int head = (int)(rr - tok * heads);
  const bool is_q = head < Hq;
  T* base = is_q ? q + tok * q_ts + (int64_t)head * q_hs : k + tok * k_ts + (int64_t)(head - Hq) * k_hs;
  const T* w = is_q ? qw : kw;
  const int d0 = 8 * l;

  float x[8], wv[8];
  load8<T>(base + d0, x);
  load8<T>(w + d0, wv);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
  ss = group_sum<G>(ss);
  const float rms = rsqrtf(ss / (float)D + p.eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] *= rms * wv[i];

  const int rope = p.rope_dim, half = rope >> 1;
  const int64_t pos = p.pos_is_i64 ? reinterpret_cast<const int64_t*>(p.positions)[tok]
                                   : (int64_t) reinterpret_cast<const int32_t*>(p.positions)[tok];
  // analytic angles: the workgroup's rows belong to a few consecutive tokens (RW rows, heads per token), and every head of a token
  // rotates by the same rope / 2 angles - one table per workgroup ((cos, sin) per token and frequency, at most RW * D / 2 = 1024
  // entries) instead of eight sincos per thread (16384 tokens x 40 heads: 142 -> see NOTEBOOK round 5 (22))
  __shared__ float2 s_cs[ANALYTIC ? 1024 : 1];
  int64_t tok_first = 0;
  if constexpr (ANALYTIC) {
    const int64_t total = tokens * heads;
    const int64_t r0 = (int64_t)blockIdx.x * RW, r1 = r0 + RW - 1 < total ? r0 + RW - 1 : total - 1;
    const bool small = total < (1ll << 32);
    tok_first = small ? (int64_t)((uint32_t)r0 / (uint32_t)heads) : r0 / heads;
    const int64_t tok_last = small ? (int64_t)((uint32_t)r1 / (uint32_t)heads) : r1 / heads;
    const int n = (int)(tok_last - tok_first + 1) * half;
    for (int i = threadIdx.x; i < n; i += 256) {
      const int t = i / half, jf = i - t * half;
      const int64_t pt = p.pos_is_i64 ? reinterpret_cast<const int64_t*>(p.positions)[tok_first + t]
                                      : (int64_t) reinterpret_cast<const int32_t*>(p.positions)[tok_first + t];
      const float theta = (float)pt * yarn_freq(p.log2_base, rope, jf, p.factor, p.low, p.high);
      float sn, cs;
      sincosf(theta, &sn, &cs);  // (one range reduction for both)
      s_cs[i] = make_float2(cs, sn);
    }
    __syncthreads();
  }
  auto angle = [&](int half_idx, float& c, float& s) {
    if constexpr (ANALYTIC) {
      const float2 v = s_cs[(int)(tok - tok_first) * half + half_idx];
      c = v.x;
      s = v.y;
    } else {
      const float* row_cs = p.cos_sin_cache + pos * rope;
      c = row_cs[half_idx];
      s = row_cs[half + half_idx];
    }
  };
  float out[8];
  // cached angles of a chunk as 16-byte loads where the layout allows (8 | half, the cache 16-byte aligned: cv / sv = cos / sin of the
  // chunk's eight (NeoX) or four (interleaved) frequencies)
  float cv[8], sv[8];
  bool vec_cs = false;
  if constexpr (!ANALYTIC) {
    vec_cs = p.cs_vec != 0 && d0 < rope;
    if (vec_cs) {
      const float* row_cs = p.cos_sin_cache + pos * rope;
      if constexpr (NEOX) {
        const int h0 = d0 < half ? d0 : d0 - half;
        const float4 c0 = *reinterpret_cast<const float4*>(row_cs + h0), c1 = *reinterpret_cast<const float4*>(row_cs + h0 + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(row_cs + half + h0), s1 = *reinterpret_cast<const float4*>(row_cs + half + h0 + 4);
        cv[0] = c0.x; cv[1] = c0.y; cv[2] = c0.z; cv[3] = c0.w; cv[4] = c1.x; cv[5] = c1.y; cv[6] = c1.z; cv[7] = c1.w;
        sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w; sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
      } else {
        const float4 c0 = *reinterpret_cast<const float4*>(row_cs + (d0 >> 1)), s0 = *reinterpret_cast<const float4*>(row_cs + half + (d0 >> 1));
        cv[0] = c0.x; cv[1] = c0.y; cv[2] = c0.z; cv[3] = c0.w;
        sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w;
      }
    }
  }
  if constexpr (NEOX) {
    // partner values: y[p] = x_raw[p] * (rms * w[p]); whole 16-byte chunks when the halves are chunk aligned
    float y2[8];
    const bool rot_chunk = d0 < rope;
    if ((half & 7) == 0) {
      const int p0 = d0 < half ? d0 + half : d0 - half;
      if (rot_chunk) {
        float xr[8], wr[8];
        load8<T>(base + p0, xr);
        load8<T>(w + p0, wr);
#pragma unroll
        for (int i = 0; i < 8; ++i) y2[i] = xr[i] * (rms * wr[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int d = d0 + i;
        const int pi = d < half ? d + half : d - half;
        y2[i] = d < rope ? (float)base[pi] * (rms * (float)w[pi]) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d = d0 + i;
      if (d < rope) {
        float c, s;
        if (vec_cs) { c = cv[i]; s = sv[i]; } else angle(d < half ? d : d - half, c, s);
        float r = x[i] * c + (d < half ? -y2[i] : y2[i]) * s;
        if constexpr (ANALYTIC) r *= p.attention_factor;
        out[i] = r;
      } else {
        out[i] = x[i];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int d = d0 + i;
      if (d < rope) {
        float c, s;
        if (vec_cs) { c = cv[i >> 1]; s = sv[i >> 1]; } else angle(d >> 1, c, s);
        float re = x[i] * c - x[i + 1] * s, im = x[i] * s + x[i + 1] * c;
        if constexpr (ANALYTIC) { re *= p.attention_factor; im *= p.attention_factor; }
        out[i] = re;
        out[i + 1] = im;
      } else {
        out[i] = x[i];
        out[i + 1] = x[i + 1];
      }
    }
  }
  if (live) store8<T>(base + d0, out);
}

template <typename T, int D>
static void launch_d(hipStream_t st, void* q, void* k, const void* qw, const void* kw, const RopeParams& p, int64_t tokens,
                     int Hq, int Hk, int64_t q_ts, int64_t q_hs, int64_t k_ts, int64_t k_hs, bool neox, bool analytic) {
  constexpr int RW = 256 / (D / 8);
  const unsigned grid = (unsigned)cdiv(tokens * (Hq + Hk), RW);
#define SGLK_GO(N, A)                                                                                          \
  qknorm_rope_kernel<T, D, N, A><<<grid, 256, 0, st>>>((T*)q, (T*)k, (const T*)qw, (const T*)kw, p, tokens, Hq, Hk, \
                                                       q_ts, q_hs, k_ts, k_hs)
  if (neox) { if (analytic) SGLK_GO(true, true); else SGLK_GO(true, false); }
  else      { if (analytic) SGLK_GO(false, true); else SGLK_GO(false, false); }
#undef SGLK_GO
}

static int launch_all(const char* op, hipStream_t st, void* q, void* k, const void* qw, const void* kw, const RopeParams& p,
                      int64_t tokens, int64_t Hq, int64_t Hk, int64_t head_dim, int64_t q_ts, int64_t q_hs, int64_t k_ts,
                      int64_t k_hs, bool neox, bool analytic, int dtype) {
  SGLK_REQUIRE(head_dim == 64 || head_dim == 128 || head_dim == 256, "Unsupported head dimension for %s: %lld", op,
               (long long)head_dim);
  SGLK_REQUIRE(p.rope_dim > 0 && p.rope_dim <= head_dim && p.rope_dim % 2 == 0,
               "%s: rope_dim must be even and in (0, head_dim], got %d", op, p.rope_dim);
  const int64_t esz = dtype == SGLK_F32 ? 4 : 2, al = 16 / esz;
  SGLK_REQUIRE((uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)qw % 16 == 0 && (uintptr_t)kw % 16 == 0 &&
                   q_ts % al == 0 && q_hs % al == 0 && k_ts % al == 0 && k_hs % al == 0,
               "%s: q / k rows and the norm weights must be 16-byte aligned", op);
  if (tokens == 0 || Hq + Hk == 0) return SGLK_OK;
#define SGLK_GO_D(T)                                                                                            \
  switch (head_dim) {                                                                                           \
    case 64: launch_d<T, 64>(st, q, k, qw, kw, p, tokens, (int)Hq, (int)Hk, q_ts, q_hs, k_ts, k_hs, neox, analytic); break;   \
    case 128: launch_d<T, 128>(st, q, k, qw, kw, p, tokens, (int)Hq, (int)Hk, q_ts, q_hs, k_ts, k_hs, neox, analytic); break; \
    default: launch_d<T, 256>(st, q, k, qw, kw, p, tokens, (int)Hq, (int)Hk, q_ts, q_hs, k_ts, k_hs, neox, analytic); break;  \
  }
  SGLK_DISPATCH_FLOAT(dtype, T, SGLK_GO_D(T))
#undef SGLK_GO_D
  return check_launch(op);
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_fused_qknorm_rope_cache(sglk_stream_t stream, void* q, void* k, const void* q_weight,
                                            const void* k_weight, const float* cos_sin_cache, const void* positions,
                                            int positions_are_int64, int64_t tokens, int64_t num_q_heads,
                                            int64_t num_k_heads, int64_t head_dim, int64_t rope_dim,
                                            int64_t q_token_stride, int64_t q_head_stride, int64_t k_token_stride,
                                            int64_t k_head_stride, int is_neox, float eps, int dtype) {
  using namespace sglk;
  RopeParams p{};
  p.cos_sin_cache = cos_sin_cache;
  p.positions = positions;
  p.pos_is_i64 = positions_are_int64;
  p.rope_dim = (int)rope_dim;
  p.eps = eps;
  p.cs_vec = (rope_dim % 16 == 0 && (uintptr_t)cos_sin_cache % 16 == 0) ? 1 : 0;
  return launch_all("fused_inplace_qknorm_rope", (hipStream_t)stream, q, k, q_weight, k_weight, p, tokens, num_q_heads,
                    num_k_heads, head_dim, q_token_stride, q_head_stride, k_token_stride, k_head_stride, is_neox != 0, false,
                    dtype);
}

extern "C" int sglk_fused_qknorm_rope_yarn(sglk_stream_t stream, void* qkv, const void* q_weight, const void* k_weight,
                                           const int32_t* position_ids, int64_t tokens, int64_t num_q_heads,
                                           int64_t num_k_heads, int64_t num_v_heads, int64_t head_dim, int64_t rotary_dim,
                                           float eps, float base, int is_neox, float factor, float low, float high,
                                           float attention_factor, int dtype) {
  using namespace sglk;
  RopeParams p{};
  p.positions = position_ids;
  p.pos_is_i64 = 0;
  p.rope_dim = (int)rotary_dim;
  p.eps = eps;
  p.log2_base = log2f(base);
  p.factor = factor;
  p.low = low;
  p.high = high;
  p.attention_factor = attention_factor;
  const int64_t row = (num_q_heads + num_k_heads + num_v_heads) * head_dim;
  const int64_t esz = dtype == SGLK_F32 ? 4 : 2;
  void* kptr = (char*)qkv + num_q_heads * head_dim * esz;
  return launch_all("fused_qk_norm_rope", (hipStream_t)stream, qkv, kptr, q_weight, k_weight, p, tokens, num_q_heads,
                    num_k_heads, head_dim, row, head_dim, row, head_dim, is_neox != 0, true, dtype);
}
