// rotary_embedding for gfx950.
//
// Replaces reference src/sycl/Rope.cpp:453-471 (entry), :328-391 (2-D in-place neox / interleaved,
// RotaryEmbeddingBatched :42-86) and :403-451 (3-D out-of-place DeepSeek variant). Semantics kept:
//   cos_sin_cache [max_pos, rot_dim] in the dtype of q: cos = first rot_dim/2 entries, sin = the rest;
//   for every head, pair (x, y) = (v[i], v[i + rot_dim/2]) (neox) or (v[2i], v[2i+1]) (interleaved),
//   i < rot_dim/2:  x' = x cos_i - y sin_i ;  y' = x sin_i + y cos_i   in fp32, rounded once to T;
//   elements past rot_dim pass through. positions are int64.
// Design: one workgroup per token, 16-byte vectors along the rotary index where the layout allows.
#include "common.h"

namespace sglk {
namespace {

template <typename T, int VEC, bool NEOX>
__global__ __launch_bounds__(256) void rope_kernel(T* q_out, T* k_out, const T* q_in, const T* k_in,
                                                   const int64_t* __restrict__ positions,
                                                   const T* __restrict__ cache, int hq, int hk, int head_size,
                                                   int rot_dim, int64_t q_tok_stride, int64_t q_head_stride,
                                                   int64_t k_tok_stride, int64_t k_head_stride,
                                                   int64_t qo_tok_stride, int64_t qo_head_stride,
                                                   int64_t ko_tok_stride, int64_t ko_head_stride, bool copy_tail) {
  const int64_t tok = blockIdx.x;
  const int embed = rot_dim / 2;
  const int64_t pos = positions[tok];
  const T* cosp = cache + pos * rot_dim;
  const T* sinp = cosp + embed;
  const int per_head = embed / VEC;
  const int total = (hq + hk) * per_head;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int head = idx / per_head;
    const int i = (idx - head * per_head) * VEC;
    const bool is_q = head < hq;
    const int h = is_q ? head : head - hq;
    const T* src = is_q ? q_in + tok * q_tok_stride + (int64_t)h * q_head_stride
                        : k_in + tok * k_tok_stride + (int64_t)h * k_head_stride;
    T* dst = is_q ? q_out + tok * qo_tok_stride + (int64_t)h * qo_head_stride
                  : k_out + tok * ko_tok_stride + (int64_t)h * ko_head_stride;
    const Vec<T, VEC> c = load_vec<T, VEC>(cosp + i);
    const Vec<T, VEC> s = load_vec<T, VEC>(sinp + i);
    if constexpr (NEOX) {
      const Vec<T, VEC> x = load_vec<T, VEC>(src + i);
      const Vec<T, VEC> y = load_vec<T, VEC>(src + embed + i);
      Vec<T, VEC> ox, oy;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xf = (float)x[e], yf = (float)y[e], cf = (float)c[e], sf = (float)s[e];
        ox[e] = (T)(xf * cf - yf * sf);
        oy[e] = (T)(xf * sf + yf * cf);
      }
      store_vec<T, VEC>(dst + i, ox);
      store_vec<T, VEC>(dst + embed + i, oy);
    } else {
      const Vec<T, 2 * VEC> v = load_vec<T, 2 * VEC>(src + 2 * i);
      Vec<T, 2 * VEC> o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xf = (float)v[2 * e], yf = (float)v[2 * e + 1], cf = (float)c[e], sf = (float)s[e];
        o[2 * e] = (T)(xf * cf - yf * sf);
        o[2 * e + 1] = (T)(xf * sf + yf * cf);
      }
      store_vec<T, 2 * VEC>(dst + 2 * i, o);
    }
  }
  if (copy_tail && rot_dim < head_size) {  // out-of-place call: carry the un-rotated tail over
    const int tail = head_size - rot_dim;
    for (int idx = threadIdx.x; idx < (hq + hk) * tail; idx += 256) {
      const int head = idx / tail, j = rot_dim + idx % tail;
      if (head < hq) q_out[tok * qo_tok_stride + (int64_t)head * qo_head_stride + j] =
          q_in[tok * q_tok_stride + (int64_t)head * q_head_stride + j];
      else k_out[tok * ko_tok_stride + (int64_t)(head - hq) * ko_head_stride + j] =
          k_in[tok * k_tok_stride + (int64_t)(head - hq) * k_head_stride + j];
    }
  }
}

}  // namespace
}  // namespace sglk

namespace sglk {
namespace {

struct RopeArgs {
  void* q_out; void* k_out; const void* q; const void* k; const int64_t* positions; const void* cache;
  int64_t tokens, num_heads, num_kv_heads, head_size, rot_dim;
  int64_t q_ts, q_hs, k_ts, k_hs, qo_ts, qo_hs, ko_ts, ko_hs;
  bool copy_tail;
};

template <typename T, int V, bool NEOX>
static void rope_launch(hipStream_t st, const RopeArgs& a) {
  rope_kernel<T, V, NEOX><<<(unsigned)a.tokens, 256, 0, st>>>(
      (T*)a.q_out, (T*)a.k_out, (const T*)a.q, (const T*)a.k, a.positions, (const T*)a.cache, (int)a.num_heads,
      (int)a.num_kv_heads, (int)a.head_size, (int)a.rot_dim, a.q_ts, a.q_hs, a.k_ts, a.k_hs, a.qo_ts, a.qo_hs, a.ko_ts,
      a.ko_hs, a.copy_tail);
}

// vector width along the rotary index: everything touched must stay aligned to the vector
template <typename T>
static bool rope_aligned(const RopeArgs& a, int v, bool neox) {
  const int64_t bytes = (int64_t)v * sizeof(T) * (neox ? 1 : 2);
  const int64_t embed = a.rot_dim / 2;
  const int64_t strides[] = {a.q_ts, a.q_hs, a.k_ts, a.k_hs, a.qo_ts, a.qo_hs, a.ko_ts, a.ko_hs};
  for (int64_t s : strides)
    if ((s * (int64_t)sizeof(T)) % bytes) return false;
  const uintptr_t ptrs[] = {(uintptr_t)a.q, (uintptr_t)a.k, (uintptr_t)a.q_out, (uintptr_t)a.k_out};
  for (uintptr_t ptr : ptrs)
    if (ptr % bytes) return false;
  if (neox && (embed * (int64_t)sizeof(T)) % bytes) return false;
  return embed % v == 0 && (uintptr_t)a.cache % (v * sizeof(T)) == 0 && (a.rot_dim % v) == 0;
}

template <typename T>
static void rope_dispatch(hipStream_t st, const RopeArgs& a, bool is_neox) {
  constexpr int kMax = 16 / sizeof(T);
  if (is_neox) {
    if (rope_aligned<T>(a, kMax, true)) rope_launch<T, kMax, true>(st, a); else rope_launch<T, 1, true>(st, a);
  } else {
    if (rope_aligned<T>(a, kMax / 2, false)) rope_launch<T, kMax / 2, false>(st, a); else rope_launch<T, 1, false>(st, a);
  }
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_rotary_embedding(sglk_stream_t stream, void* q_out, void* k_out, const void* q, const void* k,
                                     const int64_t* positions, const void* cos_sin_cache, int64_t tokens,
                                     int64_t num_heads, int64_t num_kv_heads, int64_t head_size, int64_t rot_dim,
                                     int64_t q_tok_stride, int64_t q_head_stride, int64_t k_tok_stride,
                                     int64_t k_head_stride, int64_t qo_tok_stride, int64_t qo_head_stride,
                                     int64_t ko_tok_stride, int64_t ko_head_stride, int is_neox, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size,
               "rotary_embedding: rot_dim must be even and <= head_size (got %lld, head_size %lld)", (long long)rot_dim,
               (long long)head_size);
  if (tokens == 0) return SGLK_OK;
  const RopeArgs a{q_out, k_out, q, k, positions, cos_sin_cache, tokens, num_heads, num_kv_heads, head_size, rot_dim,
                   q_tok_stride, q_head_stride, k_tok_stride, k_head_stride, qo_tok_stride, qo_head_stride,
                   ko_tok_stride, ko_head_stride, q_out != q};
  hipStream_t st = (hipStream_t)stream;
  SGLK_DISPATCH_FLOAT(dtype, T, { rope_dispatch<T>(st, a, is_neox != 0); });
  return check_launch("rotary_embedding");
}
