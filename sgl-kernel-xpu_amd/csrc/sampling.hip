// Top-k / top-p / min-p filtering and sampling from probabilities.
//
// Replaces reference src/sycl/TopKRenormProbs.cpp, TopPRenormProbs.cpp, TopKTopPSamplingFromProbs.cpp and
// MinPSamplingFromProbs.cpp (schemas src/torch_extension_sycl.cc:66-80; wrappers python/sgl_kernel/sampling.py).
// Contract (reference tests/test_sampling.py): with per-row parameters k, p (or min_p)
//   top-k keeps   x >= (k-th largest value)                                   (ties with the pivot are kept)
//   top-p keeps   the largest values whose total mass reaches p (x >= t_p, t_p = largest t with mass{x >= t} >= p)
//   min-p keeps   x >= min_p * max(x)
//   renorm ops:   out = kept ? x / sum(kept) : 0
//   sampling ops: draw an index with probability proportional to x over the rows' kept set (joint: both filters).
//
// The reference (flashinfer's algorithm) finds its pivots by rejection rounds of sampling; this build computes the
// pivots EXACTLY with a radix select over the fp32 bit patterns, so membership is exact and independent of the random
// stream, and the draw is one inverse-CDF lookup:
//   * one workgroup (1024 threads) per row; keys = the float bits (non-negative floats order like unsigned integers);
//   * three radix passes (12 + 10 + 10 bits) with LDS histograms of COUNT and MASS per digit; masses are added as
//     40-bit fixed point integers (x * 2^40), so every sum is exact and independent of the order of the atomic adds:
//     thresholds, normalisers and draws are bit-reproducible from the same (seed, offset);
//   * the first pass's histogram serves both the top-k and the top-p select (no prefix yet); 12 bits there spread
//     a softmax's few exponent values over many bins (LDS atomic conflicts);
//   * the draw: u64 from Philox4x32-10(seed; offset, row), target = floor(u * Z / 2^64) in fixed point, thread chunk
//     sums + one block scan locate the chunk, its owner walks it.
// Rows of 128k-152k fp32 probabilities (0.5 MB) are re-read from L2 by the 5-7 passes; the op is latency-bound.
#include <math.h>

#include "common.h"

namespace sglk {
namespace {

constexpr int kT = 1024;
constexpr int kBins1 = 4096;
constexpr float kFix = 1099511627776.0f;  // 2^40

struct SampleParams {
  const float* probs;       // [rows_in, V]
  float* renorm;            // renorm ops: [B, V]
  int32_t* out;             // sampling ops: [B]
  const int64_t* indices;   // optional row map (sampling ops)
  const void* k_arr;        // per-row k (int32 or int64), optional
  const float* p_arr;       // per-row p / min_p, optional
  int k_is_i64;
  int64_t k_val;
  float p_val;
  int V;
  int use_k, use_p, use_minp, do_sample;
  uint64_t seed, offset;
};

__device__ __forceinline__ uint32_t key_of(float x) { return x > 0.f ? __float_as_uint(x) : 0u; }  // (NaN, negatives -> 0)
__device__ __forceinline__ uint64_t fix_of(float x) { return x > 0.f ? (uint64_t)(x * kFix) : 0ull; }

struct Philox {
  uint32_t c[4];
  __device__ Philox(uint64_t seed, uint64_t offset, uint64_t row) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    c[0] = (uint32_t)offset; c[1] = (uint32_t)(offset >> 32); c[2] = (uint32_t)row; c[3] = (uint32_t)(row >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
      c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
  }
};

// block-wide inclusive suffix sum over threads (thread t gets sum of v over threads >= t); red: >= 16 entries
template <typename V>
__device__ __forceinline__ V block_suffix_sum(V v, V* red, V& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  V s = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const V t = __shfl_down(s, o, 64);
    if (lane + o < 64) s += t;
  }
  __syncthreads();
  if (lane == 0) red[w] = s;  // wave total
  __syncthreads();
  V after = 0, tot = 0;
  for (int i = 0; i < kT / 64; ++i) {
    const V r = red[i];
    tot += r;
    if (i > w) after += r;
  }
  total = tot;
  return s + after;
}

struct Shared {
  uint32_t cnt[kBins1];
  unsigned long long mass[kBins1];
  unsigned long long red64[16];
  uint32_t red32[16];
  // select state: [0] = count criterion, [1] = mass criterion
  uint32_t prefix[2];
  unsigned long long above[2];
  int exhausted[2];
  unsigned long long pick_above;
  uint32_t pick_digit;
  int pick_found;
  float fmax_;
  int out_idx;
};

// Histogram of digit (key >> shift) & (nb - 1) over the row's elements whose bits above (shift + bits) equal `prefix`
__device__ void build_hist(Shared& s, const float* row, int V, uint32_t prefix, int shift, int bits, bool match_all) {
  const int nb = 1 << bits;
  for (int i = threadIdx.x; i < nb; i += kT) { s.cnt[i] = 0; s.mass[i] = 0ull; }
  __syncthreads();
  for (int i = threadIdx.x; i < V; i += kT) {
    const float x = row[i];
    const uint32_t key = key_of(x);
    if (match_all || (key >> (shift + bits)) == (prefix >> (shift + bits))) {
      const uint32_t d = (key >> shift) & (uint32_t)(nb - 1);
      atomicAdd(&s.cnt[d], 1u);
      atomicAdd(&s.mass[d], fix_of(x));
    }
  }
  __syncthreads();
}

// From the histogram: the largest digit b with above + sum_{b' >= b} h[b'] >= target; updates the criterion's state.
__device__ void pick_digit(Shared& s, int crit, unsigned long long target, int shift, int bits) {
  const int nb = 1 << bits, per = nb / kT > 0 ? nb / kT : 1;  // bins per thread (4 or 1)
  const int b0 = threadIdx.x * per;
  unsigned long long loc[4] = {0, 0, 0, 0}, mine = 0;
  if (b0 < nb) {
    for (int j = 0; j < per; ++j) {
      loc[j] = crit == 0 ? (unsigned long long)s.cnt[b0 + j] : s.mass[b0 + j];
      mine += loc[j];
    }
  }
  unsigned long long total;
  const unsigned long long suf = block_suffix_sum<unsigned long long>(mine, s.red64, total);  // bins >= b0
  if (threadIdx.x == 0) s.pick_found = 0;
  __syncthreads();
  const unsigned long long above = s.above[crit];
  if (b0 < nb) {
    // cum(b) for my bins, descending: cum(b0 + per - 1) = above + suf - (sum of my lower bins)
    unsigned long long higher = above + suf - mine;  // bins strictly above my range
    for (int j = per - 1; j >= 0; --j) {
      const unsigned long long cum = higher + loc[j];  // includes bin b0 + j
      if (cum >= target && higher < target) {          // the boundary bin
        s.pick_digit = (uint32_t)(b0 + j);
        s.pick_above = higher;
        s.pick_found = 1;
      }
      higher = cum;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s.pick_found) {
      s.prefix[crit] |= s.pick_digit << shift;
      s.above[crit] = s.pick_above;
    } else {
      s.exhausted[crit] = 1;  // fewer elements / less mass than asked for: keep everything under this prefix
    }
  }
  __syncthreads();
}

template <int DUMMY>
__global__ __launch_bounds__(kT) void sampling_kernel(SampleParams p) {
  __shared__ Shared s;
  const int b = blockIdx.x;
  const int V = p.V;
  const int64_t src_row = p.indices ? p.indices[b] : b;
  const float* row = p.probs + src_row * (int64_t)V;

  uint32_t thr = 0;  // keep key >= thr
  if (p.use_minp) {
    float m = 0.f;
    for (int i = threadIdx.x; i < V; i += kT) m = fmaxf(m, row[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s.red32[threadIdx.x >> 6] = __float_as_uint(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      float mm = 0.f;
      for (int i = 0; i < kT / 64; ++i) mm = fmaxf(mm, __uint_as_float(s.red32[i]));
      s.fmax_ = mm;
    }
    __syncthreads();
    const float mp = p.p_arr ? p.p_arr[b] : p.p_val;
    thr = key_of(mp * s.fmax_);
  } else {
    unsigned long long target[2] = {0, 0};
    bool use[2] = {p.use_k != 0, p.use_p != 0};
    if (use[0]) {
      long long k = p.k_arr ? (p.k_is_i64 ? reinterpret_cast<const int64_t*>(p.k_arr)[b] : (long long)reinterpret_cast<const int32_t*>(p.k_arr)[b])
                            : (long long)p.k_val;
      if (k > V) k = V;
      if (k <= 0) use[0] = false;  // (nothing asked: no top-k filter)
      target[0] = (unsigned long long)k;
    }
    if (use[1]) {
      const float pp = p.p_arr ? p.p_arr[b] : p.p_val;
      target[1] = pp >= 1.0f ? ~0ull : (unsigned long long)((double)pp * (double)kFix);
      if (pp >= 1.0f) use[1] = false;  // p = 1: everything is kept (the row sum may fall short of 1 by rounding)
    }
    if (threadIdx.x == 0) {
      s.prefix[0] = s.prefix[1] = 0;
      s.above[0] = s.above[1] = 0;
      s.exhausted[0] = s.exhausted[1] = 0;
    }
    __syncthreads();
    const int shifts[3] = {20, 10, 0}, bitsv[3] = {12, 10, 10};
    for (int pass = 0; pass < 3; ++pass) {
      if (pass == 0) {
        if (use[0] || use[1]) build_hist(s, row, V, 0, shifts[0], bitsv[0], true);
        for (int c = 0; c < 2; ++c)
          if (use[c]) pick_digit(s, c, target[c], shifts[0], bitsv[0]);
      } else {
        for (int c = 0; c < 2; ++c) {
          if (use[c] && !s.exhausted[c]) {
            build_hist(s, row, V, s.prefix[c], shifts[pass], bitsv[pass], false);
            pick_digit(s, c, target[c], shifts[pass], bitsv[pass]);
          }
        }
      }
    }
    const uint32_t tk = use[0] ? s.prefix[0] : 0u, tp = use[1] ? s.prefix[1] : 0u;
    thr = tk > tp ? tk : tp;
  }

  // ---- normaliser (and per-thread chunk sums for the draw): thread t owns elements [t * C, (t + 1) * C)
  const int C = (V + kT - 1) / kT;
  const int lo = threadIdx.x * C, hi = lo + C < V ? lo + C : V;
  unsigned long long mine = 0;
  for (int i = lo; i < hi; ++i) {
    const float x = row[i];
    if (key_of(x) >= thr) mine += fix_of(x);
  }
  unsigned long long Z;
  const unsigned long long suf = block_suffix_sum<unsigned long long>(mine, s.red64, Z);
  if (!p.do_sample) {
    const float inv = Z > 0 ? (float)((double)kFix / (double)Z) : 0.f;
    float* orow = p.renorm + (int64_t)b * V;
    for (int i = threadIdx.x; i < V; i += kT) {
      const float x = row[i];
      orow[i] = key_of(x) >= thr ? x * inv : 0.f;
    }
    return;
  }
  if (threadIdx.x == 0) s.out_idx = -1;
  __syncthreads();
  const Philox rng(p.seed, p.offset, (uint64_t)b);
  const unsigned long long u = ((unsigned long long)rng.c[0] << 32) | rng.c[1];
  const unsigned long long target = __umul64hi(u, Z);  // in [0, Z)
  const unsigned long long before = Z - suf;            // mass of the threads before me
  if (Z > 0 && mine > 0 && before <= target && target < before + mine) {
    unsigned long long acc = before;
    int pickd = hi - 1;
    for (int i = lo; i < hi; ++i) {
      const float x = row[i];
      if (key_of(x) >= thr) {
        acc += fix_of(x);
        if (acc > target) { pickd = i; break; }
      }
    }
    s.out_idx = pickd;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int r = s.out_idx;
    if (r < 0) r = 0;  // an all-zero row has no mass to draw from
    p.out[b] = r;
  }
}

static int run(hipStream_t st, const SampleParams& p, int64_t batch, const char* op) {
  SGLK_REQUIRE(p.V > 0, "%s: vocab size must be positive", op);
  if (batch == 0) return SGLK_OK;
  sampling_kernel<0><<<(unsigned)batch, kT, 0, st>>>(p);
  return check_launch(op);
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_top_k_renorm_probs(sglk_stream_t stream, float* renorm_probs, const float* probs, const int64_t* top_k_arr,
                                       int64_t top_k_val, int64_t batch, int64_t vocab) {
  using namespace sglk;
  SampleParams p{};
  p.probs = probs; p.renorm = renorm_probs; p.k_arr = top_k_arr; p.k_is_i64 = 1; p.k_val = top_k_val; p.V = (int)vocab; p.use_k = 1;
  return run((hipStream_t)stream, p, batch, "top_k_renorm_probs");
}

extern "C" int sglk_top_p_renorm_probs(sglk_stream_t stream, float* renorm_probs, const float* probs, const float* top_p_arr,
                                       float top_p_val, int64_t batch, int64_t vocab) {
  using namespace sglk;
  SampleParams p{};
  p.probs = probs; p.renorm = renorm_probs; p.p_arr = top_p_arr; p.p_val = top_p_val; p.V = (int)vocab; p.use_p = 1;
  return run((hipStream_t)stream, p, batch, "top_p_renorm_probs");
}

extern "C" int sglk_top_k_top_p_sampling_from_probs(sglk_stream_t stream, int32_t* output, const float* probs,
                                                    const int64_t* indices, const int32_t* top_k_arr, int64_t top_k_val,
                                                    const float* top_p_arr, float top_p_val, int use_top_k, int64_t batch,
                                                    int64_t vocab, uint64_t philox_seed, uint64_t philox_offset) {
  using namespace sglk;
  SampleParams p{};
  p.probs = probs; p.out = output; p.indices = indices; p.k_arr = top_k_arr; p.k_is_i64 = 0; p.k_val = top_k_val;
  p.p_arr = top_p_arr; p.p_val = top_p_val; p.V = (int)vocab; p.use_k = use_top_k; p.use_p = 1; p.do_sample = 1;
  p.seed = philox_seed; p.offset = philox_offset;
  return run((hipStream_t)stream, p, batch, use_top_k ? "top_k_top_p_sampling_from_probs" : "top_p_sampling_from_probs");
}

extern "C" int sglk_min_p_sampling_from_probs(sglk_stream_t stream, int32_t* output, const float* probs, const int64_t* indices,
                                              const float* min_p_arr, float min_p_val, int64_t batch, int64_t vocab,
                                              uint64_t philox_seed, uint64_t philox_offset) {
  using namespace sglk;
  SampleParams p{};
  p.probs = probs; p.out = output; p.indices = indices; p.p_arr = min_p_arr; p.p_val = min_p_val; p.V = (int)vocab;
  p.use_minp = 1; p.do_sample = 1; p.seed = philox_seed; p.offset = philox_offset;
  return run((hipStream_t)stream, p, batch, "min_p_sampling_from_probs");
}
