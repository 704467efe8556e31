// Attention prologue / epilogue passes around `fwd`: merging two partial attention states and
// writing new K/V rows into the flat KV cache. Both are single HBM passes.
//
// ---- merge_state / merge_state_v2 ----
// Replaces reference src/sycl/merge_states.cpp:138-361 (schemas src/torch_extension_sycl.cc:232-234). Two partial
// results (v_a, s_a), (v_b, s_b) of softmax attention over disjoint key sets - v [tokens, heads, d] normalised
// outputs, s [tokens, heads] fp32 log-sum-exp - become the result over the union:
//   m = max(s_a, s_b);  w_a = e^(s_a - m), w_b = e^(s_b - m);  z = max(w_a + w_b, FLT_MIN)
//   v = T(v_a * (w_a / z) + v_b * (w_b / z));   s = log(z) + m
// merge_state_v2 works in base e (MergePrefixSuffix, merge_states.cpp:37-137), merge_state in base 2 (MergeState,
// :173-272: the flashinfer convention). A non-finite s (+inf marks "no keys seen" in the reference's tests, NaN
// likewise) counts as -inf: that side gets weight 0 (:94-95, :231-232).
// Kernel: 16 bytes of v per lane, the (token, head) of a lane derived from its flat index; the two scalars of a head
// are re-read by each of its d/8 lanes (one cache line per 16 heads). 3 x tokens x heads x d x sizeof(T) bytes of HBM.
//
// ---- store_cache ----
// Replaces reference src/sycl/KVCache.cpp:11-160 (schema src/torch_extension_sycl.cc:122-125): row t of k / v
// (row stride given: a per-head slice of a wider tensor is addressed in place) is copied to row indices[t] of the
// dense k_cache / v_cache; a negative index skips the token. A pure byte copy: bit-exact for every dtype.
#include <limits>

#include "common.h"

namespace sglk {
namespace {

template <typename T, bool BASE2>
__global__ __launch_bounds__(256) void merge_state_kernel(T* __restrict__ v_out, float* __restrict__ s_out,
                                                          const T* __restrict__ v_a, const float* __restrict__ s_a,
                                                          const T* __restrict__ v_b, const float* __restrict__ s_b,
                                                          int64_t total_packs, int packs_per_head) {
  constexpr int kPack = 16 / sizeof(T);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_packs) return;
  const int64_t th = idx / packs_per_head;  // token * heads + head
  const int pack = (int)(idx - th * packs_per_head);

  float sa = s_a[th], sb = s_b[th];
  const float ninf = -std::numeric_limits<float>::infinity();
  sa = __builtin_isfinite(sa) ? sa : ninf;
  sb = __builtin_isfinite(sb) ? sb : ninf;
  const float m = fmaxf(sa, sb);
  const float wa = BASE2 ? exp2f(sa - m) : expf(sa - m);
  const float wb = BASE2 ? exp2f(sb - m) : expf(sb - m);
  const float z = fmaxf(wa + wb, std::numeric_limits<float>::min());
  const float ca = wa / z, cb = wb / z;

  const int64_t off = idx * kPack;
  const Vec<T, kPack> a = load_vec<T, kPack>(v_a + off), b = load_vec<T, kPack>(v_b + off);
  Vec<T, kPack> o;
#pragma unroll
  for (int i = 0; i < kPack; ++i) o[i] = (T)((float)a[i] * ca + (float)b[i] * cb);
  store_vec<T, kPack>(v_out + off, o);
  if (s_out != nullptr && pack == 0) s_out[th] = (BASE2 ? log2f(z) : logf(z)) + m;
}

// one workgroup of 64 * WAVES lanes per RPB token rows; V = bytes per lane and access
template <int V>
__global__ __launch_bounds__(256) void store_cache_kernel(char* __restrict__ k_cache, char* __restrict__ v_cache,
                                                          const char* __restrict__ k, const char* __restrict__ v,
                                                          const int64_t* __restrict__ indices, int64_t tokens,
                                                          int64_t row_bytes, int64_t k_stride_bytes,
                                                          int64_t v_stride_bytes, int lanes_per_row) {
  struct alignas(V) Pack { char b[V]; };
  const int rows_per_block = blockDim.x / lanes_per_row;
  const int sub = threadIdx.x / lanes_per_row, l = threadIdx.x - sub * lanes_per_row;
  const int64_t t = (int64_t)blockIdx.x * rows_per_block + sub;
  if (sub >= rows_per_block || t >= tokens) return;
  const int64_t slot = indices[t];
  if (slot < 0) return;
  const char* ks = k + t * k_stride_bytes;
  const char* vs = v + t * v_stride_bytes;
  char* kd = k_cache + slot * row_bytes;
  char* vd = v_cache + slot * row_bytes;
  for (int64_t o = (int64_t)l * V; o < row_bytes; o += (int64_t)lanes_per_row * V) {
    const Pack kp = *reinterpret_cast<const Pack*>(ks + o), vp = *reinterpret_cast<const Pack*>(vs + o);
    *reinterpret_cast<Pack*>(kd + o) = kp;
    *reinterpret_cast<Pack*>(vd + o) = vp;
  }
}

template <int V>
static void launch_store(hipStream_t st, void* kc, void* vc, const void* k, const void* v, const int64_t* idx,
                         int64_t tokens, int64_t row_bytes, int64_t ks, int64_t vs) {
  int64_t lanes = cdiv(row_bytes, V);
  int lpr = 1;
  while (lpr < 256 && lpr < lanes) lpr <<= 1;  // power of two <= 256: whole rows per workgroup
  const int rows_per_block = 256 / lpr;
  store_cache_kernel<V><<<(unsigned)cdiv(tokens, rows_per_block), 256, 0, st>>>((char*)kc, (char*)vc, (const char*)k,
                                                                             (const char*)v, idx, tokens, row_bytes, ks,
                                                                             vs, lpr);
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_merge_state(sglk_stream_t stream, void* v_merged, float* s_merged, const void* v_a,
                                const float* s_a, const void* v_b, const float* s_b, int64_t tokens, int64_t heads,
                                int64_t head_size, int dtype, int base2) {
  using namespace sglk;
  const char* op = base2 ? "merge_state" : "merge_state_v2";
  SGLK_REQUIRE(tokens >= 0 && heads > 0 && head_size > 0, "%s: bad shape", op);
  const int64_t esz = dtype == SGLK_F32 ? 4 : 2;
  SGLK_REQUIRE(dtype == SGLK_F32 || dtype == SGLK_F16 || dtype == SGLK_BF16, "Unsupported dtype for %s", op);
  const int64_t pack = 16 / esz;
  SGLK_REQUIRE(head_size % pack == 0, "%s: head_size must be multiple of pack_size:%lld", op, (long long)pack);
  SGLK_REQUIRE((uintptr_t)v_a % 16 == 0 && (uintptr_t)v_b % 16 == 0 && (uintptr_t)v_merged % 16 == 0,
               "%s: v tensors must be 16-byte aligned", op);
  if (tokens == 0) return SGLK_OK;
  const int64_t total = tokens * heads * (head_size / pack);
  const unsigned grid = (unsigned)cdiv(total, 256);
  hipStream_t st = (hipStream_t)stream;
  const int pph = (int)(head_size / pack);
#define SGLK_GO(T)                                                                                              \
  if (base2)                                                                                                    \
    merge_state_kernel<T, true><<<grid, 256, 0, st>>>((T*)v_merged, s_merged, (const T*)v_a, s_a, (const T*)v_b, \
                                                      s_b, total, pph);                                         \
  else                                                                                                          \
    merge_state_kernel<T, false><<<grid, 256, 0, st>>>((T*)v_merged, s_merged, (const T*)v_a, s_a, (const T*)v_b, \
                                                       s_b, total, pph);
  SGLK_DISPATCH_FLOAT(dtype, T, SGLK_GO(T))
#undef SGLK_GO
  return check_launch(op);
}

extern "C" int sglk_store_cache(sglk_stream_t stream, void* k_cache, void* v_cache, const void* k, const void* v,
                                const int64_t* indices, int64_t tokens, int64_t row_bytes, int64_t k_row_stride_bytes,
                                int64_t v_row_stride_bytes) {
  using namespace sglk;
  SGLK_REQUIRE(tokens >= 0 && row_bytes > 0, "store_cache: bad shape");
  SGLK_REQUIRE(k_row_stride_bytes >= 0 && v_row_stride_bytes >= 0, "store_cache: row strides must not be negative");
  if (tokens == 0) return SGLK_OK;
  // widest access every row base of all four tensors is aligned to
  const uint64_t mix = (uint64_t)(uintptr_t)k_cache | (uint64_t)(uintptr_t)v_cache | (uint64_t)(uintptr_t)k |
                       (uint64_t)(uintptr_t)v | (uint64_t)row_bytes | (uint64_t)k_row_stride_bytes |
                       (uint64_t)v_row_stride_bytes;
  hipStream_t st = (hipStream_t)stream;
  if (mix % 16 == 0) launch_store<16>(st, k_cache, v_cache, k, v, indices, tokens, row_bytes, k_row_stride_bytes, v_row_stride_bytes);
  else if (mix % 8 == 0) launch_store<8>(st, k_cache, v_cache, k, v, indices, tokens, row_bytes, k_row_stride_bytes, v_row_stride_bytes);
  else if (mix % 4 == 0) launch_store<4>(st, k_cache, v_cache, k, v, indices, tokens, row_bytes, k_row_stride_bytes, v_row_stride_bytes);
  else if (mix % 2 == 0) launch_store<2>(st, k_cache, v_cache, k, v, indices, tokens, row_bytes, k_row_stride_bytes, v_row_stride_bytes);
  else launch_store<1>(st, k_cache, v_cache, k, v, indices, tokens, row_bytes, k_row_stride_bytes, v_row_stride_bytes);
  return check_launch("store_cache");
}
