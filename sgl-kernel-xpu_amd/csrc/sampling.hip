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
//     the waves own consecutive ranges of the row: wave totals locate the wave, the masses of its runs of 64 quads + one block scan
//     locate the run, one wave scans it.
// Rows of 128k-152k fp32 probabilities (0.5 MB) are re-read from L2 by 2 (min-p) to 5 (top-k renorm) passes, each with eight 16-byte
// loads per thread in flight; the two criteria of the joint filter share every pass (count and mass histograms of one read).
#include <math.h>

#include "common.h"

namespace sglk {
namespace {

constexpr int kT = 1024;
constexpr int kBins1 = 4096;
constexpr float kFix = 1099511627776.0f;  // 2^40

struct RowWs;
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
  struct RowWs* ws;           // cluster form (several workgroups per row, one launch per pass): per-row scratch, zeroed by run()
  int slices;                 // workgroups per row there
  const int64_t* seed_ptr;    // generator state resident on the device (a launch recorded into a HIP graph): seed = *seed_ptr,
  const int64_t* offset_ptr;  // offset = *offset_ptr + offset; NULL: the scalars above
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

constexpr int kU = 8;         // 16-byte loads a thread has in flight per step of a row pass
constexpr int kBlocks = 1024;  // draw: the row is cut into <= 1024 runs of quads whose masses one block scan orders

struct Shared {
  uint32_t cnt[kBins1];
  unsigned long long mass[kBins1];
  unsigned long long bsum[kBlocks];
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
  int pick_block;
  unsigned long long pick_before;
};

// A row as aligned 16-byte quads: quad qi holds the row's elements 4 * qi - mis + {0, 1, 2, 3}; the first and the last quad may
// reach up to three floats outside the row (never outside their own aligned 16 bytes, which hold at least one row element).
// (qb .. nq: the quads this workgroup owns - the whole row, or one of `slices` runs of it in the cluster form)
struct RowQuads {
  const float4* q;
  int mis, V, nq, qb;
  __device__ RowQuads(const float* row, int V_, int slice = 0, int slices = 1) {
    mis = (int)((reinterpret_cast<uintptr_t>(row) >> 2) & 3);
    q = reinterpret_cast<const float4*>(row - mis);
    V = V_;
    const int all = (V_ + mis + 3) >> 2;
    const int per = (((all + slices - 1) / slices + 63) >> 6) << 6;
    qb = slice * per < all ? slice * per : all;
    nq = qb + per < all ? qb + per : all;
  }
};

// Scratch of one row in the cluster form: the histograms of the three passes summed over the row's workgroups (bins 0 .. 4095 pass
// 0, 4096 .. 5119 pass 1, 5120 .. 6143 pass 2), the select state after each pick, the slices' kept masses.
struct RowWs {
  unsigned long long mass[6144];
  uint32_t cnt[6144];
  unsigned long long wsum[32];
  unsigned long long above[2];
  uint32_t prefix[2];
  int exhausted[2];
  uint32_t fmax_bits, thr;
};

// f(qi, i0, x[4]) for every quad of the row (i0 = the row index of x[0]; may be < 0 or i0 + j >= V at the two ends), kU loads per
// thread issued before the first is used: a pass is bound by the CU's L2 read rate, not by kU-fold load latency.
template <typename F>
__device__ __forceinline__ void scan_quads(const RowQuads& r, F&& f) {
  for (int q0 = r.qb; q0 < r.nq; q0 += kU * kT) {
    float4 v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int qi = q0 + u * kT + (int)threadIdx.x;
      v[u] = r.q[qi < r.nq ? qi : 0];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int qi = q0 + u * kT + (int)threadIdx.x;
      const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      f(qi, qi * 4 - r.mis, x);
    }
  }
}

// f(i, x) for every element of the row (order unspecified)
template <typename F>
__device__ __forceinline__ void scan_row(const RowQuads& r, F&& f) {
  scan_quads(r, [&](int qi, int i0, const float (&x)[4]) {
    if (qi >= r.nq) return;
    if (i0 >= 0 && i0 + 4 <= r.V) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f(i0 + j, x[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((unsigned)(i0 + j) < (unsigned)r.V) f(i0 + j, x[j]);
    }
  });
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One pass over the row for one radix digit: the COUNT histogram of the count criterion's candidates (elements under its prefix) and
// the MASS histogram of the mass criterion's, both in the same read of the row (pass 0: no prefix yet, every element is a candidate).
// A single CU's VALU bounds these passes (4 cycles per wave instruction: ~1000 cycles per instruction and element of a 128k row), so
// the common case is decided per quad in a few instructions: zeros (a row that went through a top-k renorm is nearly all zeros)
// carry no mass and can only be the count criterion's pivot when the positive elements run out - which the select reports as
// `exhausted`, with the same outcome: keep everything; under a prefix, candidates are a range of the raw float bits.
__device__ void build_hist(Shared& s, const RowQuads& r, bool act0, bool act1, uint32_t pre0, uint32_t pre1, int shift, int bits,
                           bool match_all) {
  const int nb = 1 << bits;
  for (int i = threadIdx.x; i < nb; i += kT) { s.cnt[i] = 0; s.mass[i] = 0ull; }
  __syncthreads();
  const int hs = shift + bits;  // (pass 0: hs = 32, match_all)
  const uint32_t span = match_all ? 0u : 1u << hs;
  const uint32_t lo0 = match_all ? 0u : (pre0 >> hs) << hs, lo1 = match_all ? 0u : (pre1 >> hs) << hs;
  auto add = [&](float x, bool c0, bool c1) {
    const uint32_t key = key_of(x);
    if (key == 0) return;
    const uint32_t d = (key >> shift) & (uint32_t)(nb - 1);
    if (c0) atomicAdd(&s.cnt[d], 1u);
    if (c1) atomicAdd(&s.mass[d], fix_of(x));
  };
  scan_quads(r, [&](int qi, int i0, const float (&xin)[4]) {
    if (qi >= r.nq) return;
    float x[4] = {xin[0], xin[1], xin[2], xin[3]};
    if (i0 < 0 || i0 + 4 > r.V) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((unsigned)(i0 + j) >= (unsigned)r.V) x[j] = 0.f;
    }
    if (match_all) {
      if (!(x[0] > 0.f || x[1] > 0.f || x[2] > 0.f || x[3] > 0.f)) return;
#pragma unroll
      for (int j = 0; j < 4; ++j) add(x[j], act0, act1);
    } else {
      bool c0[4], c1[4], any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t bts = __float_as_uint(x[j]);  // (negative floats: bits >= 2^31, never inside [lo, lo + span))
        c0[j] = act0 && bts - lo0 < span;
        c1[j] = act1 && bts - lo1 < span;
        any = any || c0[j] || c1[j];
      }
      if (!any) return;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0[j] || c1[j]) add(x[j], c0[j], c1[j]);
    }
  });
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

// global <-> LDS traffic of the cluster form: a workgroup's histogram of pass `pass` added into the row's, and the row's read back
__device__ void flush_hist(Shared& s, RowWs* ws, int pass) {
  const int nb = pass == 0 ? 4096 : 1024, off = pass == 0 ? 0 : pass == 1 ? 4096 : 5120;
  for (int i = threadIdx.x; i < nb; i += kT) {
    if (s.cnt[i]) atomicAdd(&ws->cnt[off + i], s.cnt[i]);
    if (s.mass[i]) atomicAdd(&ws->mass[off + i], s.mass[i]);
  }
}
__device__ void load_hist(Shared& s, const RowWs* ws, int pass) {
  const int nb = pass == 0 ? 4096 : 1024, off = pass == 0 ? 0 : pass == 1 ? 4096 : 5120;
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += kT) { s.cnt[i] = ws->cnt[off + i]; s.mass[i] = ws->mass[off + i]; }
  __syncthreads();
}

// STAGE 0: the whole op, one workgroup per row (grid = rows). STAGE 1 .. 5: the cluster form for a few rows of a long vocabulary
// (grid = (slices, rows); a launch per pass, kernel boundaries are the barriers between a row's workgroups; every workgroup of a row
// repeats the - deterministic - pick from the row's summed histogram, workgroup 0 records the state for the next launch):
//   1: histogram of pass 0 (min-p: the row maximum)     2: pick 0, histogram 1     3: pick 1, histogram 2
//   4: pick 2 -> threshold; the slice's kept mass        5: normaliser from the slices' masses; renorm write or the draw
// All sums are integers: the cluster form returns the bits of the one-workgroup form.
template <int STAGE>
__global__ __launch_bounds__(kT) void sampling_kernel(SampleParams p) {
  __shared__ Shared s;
  const int b = STAGE == 0 ? blockIdx.x : blockIdx.y;
  const int slice = STAGE == 0 ? 0 : blockIdx.x, slices = STAGE == 0 ? 1 : p.slices;
  const int V = p.V;
  const int64_t src_row = p.indices ? p.indices[b] : b;
  const float* row = p.probs + src_row * (int64_t)V;
  const RowQuads r(row, V, slice, slices);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RowWs* ws = STAGE == 0 ? nullptr : p.ws + b;

  uint32_t thr = 0;  // keep key >= thr
  if (p.use_minp) {
    if constexpr (STAGE == 2 || STAGE == 3) return;
    if constexpr (STAGE <= 1) {
      float m = 0.f;
      scan_row(r, [&](int, float x) { m = fmaxf(m, x); });
      m = wave_max(m);
      if (lane == 0) s.red32[wave] = __float_as_uint(m);
      __syncthreads();
      if (threadIdx.x == 0) {
        float mm = 0.f;
        for (int i = 0; i < kT / 64; ++i) mm = fmaxf(mm, __uint_as_float(s.red32[i]));
        s.fmax_ = mm;
        if constexpr (STAGE == 1) atomicMax(&ws->fmax_bits, __float_as_uint(mm));  // (non-negative floats order like their bits)
      }
      __syncthreads();
      if constexpr (STAGE == 1) return;
    }
    if constexpr (STAGE == 5) {
      thr = ws->thr;
    } else {
      const float fmax_row = STAGE == 4 ? __uint_as_float(ws->fmax_bits) : s.fmax_;
      const float mp = p.p_arr ? p.p_arr[b] : p.p_val;
      thr = key_of(mp * fmax_row);
      if (STAGE == 4 && slice == 0 && threadIdx.x == 0) ws->thr = thr;
    }
  } else if constexpr (STAGE == 5) {
    thr = ws->thr;
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
      // (cluster form: the state the previous launch's workgroup 0 recorded; all zero before the first pick)
      s.prefix[0] = STAGE >= 2 ? ws->prefix[0] : 0;  s.prefix[1] = STAGE >= 2 ? ws->prefix[1] : 0;
      s.above[0] = STAGE >= 2 ? ws->above[0] : 0;    s.above[1] = STAGE >= 2 ? ws->above[1] : 0;
      s.exhausted[0] = STAGE >= 2 ? ws->exhausted[0] : 0;  s.exhausted[1] = STAGE >= 2 ? ws->exhausted[1] : 0;
    }
    __syncthreads();
    const int shifts[3] = {20, 10, 0}, bitsv[3] = {12, 10, 10};
    if constexpr (STAGE == 0) {
      for (int pass = 0; pass < 3; ++pass) {
        const bool act0 = use[0] && !s.exhausted[0], act1 = use[1] && !s.exhausted[1];  // (uniform: written before a barrier)
        if (!act0 && !act1) break;
        const uint32_t pre0 = s.prefix[0], pre1 = s.prefix[1];
        build_hist(s, r, act0, act1, pre0, pre1, shifts[pass], bitsv[pass], pass == 0);
        if (act0) pick_digit(s, 0, target[0], shifts[pass], bitsv[pass]);
        if (act1) pick_digit(s, 1, target[1], shifts[pass], bitsv[pass]);
      }
    } else {
      if constexpr (STAGE >= 2) {  // the pick of pass STAGE - 2 from the row's summed histogram
        constexpr int pk = STAGE - 2;
        const bool a0 = use[0] && !s.exhausted[0], a1 = use[1] && !s.exhausted[1];
        if (a0 || a1) {
          load_hist(s, ws, pk);
          if (a0) pick_digit(s, 0, target[0], shifts[pk], bitsv[pk]);
          if (a1) pick_digit(s, 1, target[1], shifts[pk], bitsv[pk]);
          if (slice == 0 && threadIdx.x == 0) {
            ws->prefix[0] = s.prefix[0];  ws->prefix[1] = s.prefix[1];
            ws->above[0] = s.above[0];    ws->above[1] = s.above[1];
            ws->exhausted[0] = s.exhausted[0];  ws->exhausted[1] = s.exhausted[1];
          }
        }
      }
      if constexpr (STAGE <= 3) {  // this workgroup's share of the histogram of pass STAGE - 1
        constexpr int ps = STAGE - 1;
        const bool act0 = use[0] && !s.exhausted[0], act1 = use[1] && !s.exhausted[1];
        if (act0 || act1) {
          const uint32_t pre0 = s.prefix[0], pre1 = s.prefix[1];
          build_hist(s, r, act0, act1, pre0, pre1, shifts[ps], bitsv[ps], ps == 0);
          flush_hist(s, ws, ps);
        }
        return;
      }
    }
    const uint32_t tk = use[0] ? s.prefix[0] : 0u, tp = use[1] ? s.prefix[1] : 0u;
    thr = tk > tp ? tk : tp;
    if (STAGE == 4 && slice == 0 && threadIdx.x == 0) ws->thr = thr;
  }

  // ---- normaliser: wave w owns the quads [w * QW, (w + 1) * QW) (row order = wave order), a thread adds up its own quads' kept mass
  // (no cross-lane traffic inside the pass), one wave sum at the end
  const int QW = ((((r.nq - r.qb + kT / 64 - 1) / (kT / 64)) + 63) >> 6) << 6;
  auto kept_mass = [&](int qi, int qend, const float4& v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    const int i0 = qi * 4 - r.mis;
    unsigned long long m4 = 0;
    if (qi < qend) {
      // (as signed integers the bits of a positive float order like its key; negatives are below every threshold; fix_of(NaN) = 0)
      const int t = (int)thr;
      const bool k0 = __float_as_int(x[0]) >= t, k1 = __float_as_int(x[1]) >= t, k2 = __float_as_int(x[2]) >= t,
                 k3 = __float_as_int(x[3]) >= t;
      if (k0 || k1 || k2 || k3) {
        const bool full = i0 >= 0 && i0 + 4 <= V;
        const bool k[4] = {k0, k1, k2, k3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k[j] && (full || (unsigned)(i0 + j) < (unsigned)V)) m4 += fix_of(x[j]);
      }
    }
    return m4;
  };
  {
    const int wq0 = r.qb + wave * QW < r.nq ? r.qb + wave * QW : r.nq, wq1 = wq0 + QW < r.nq ? wq0 + QW : r.nq;
    unsigned long long mine = 0;
    for (int q0 = wq0; q0 < wq1; q0 += kU * 64) {
      float4 v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int qi = q0 + u * 64 + lane;
        v[u] = r.q[qi < wq1 ? qi : 0];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) mine += kept_mass(q0 + u * 64 + lane, wq1, v[u]);
    }
    const unsigned long long wt = wave_sum_u64(mine);
    __syncthreads();
    if (lane == 0) s.red64[wave] = wt;
    __syncthreads();
  }
  unsigned long long Z = 0;
  for (int i = 0; i < kT / 64; ++i) Z += s.red64[i];
  unsigned long long slice_before = 0;  // (cluster form) kept mass of the slices in front of mine
  if constexpr (STAGE == 4) {
    if (threadIdx.x == 0) ws->wsum[slice] = Z;
    return;
  }
  if constexpr (STAGE == 5) {
    const unsigned long long mine_z = Z;
    Z = 0;
    for (int i = 0; i < slices; ++i) {
      const unsigned long long t = ws->wsum[i];
      if (i < slice) slice_before += t;
      Z += t;
    }
    (void)mine_z;
  }
  if (!p.do_sample) {
    const float inv = Z > 0 ? (float)((double)kFix / (double)Z) : 0.f;
    float* orow = p.renorm + (int64_t)b * V;
    if (((reinterpret_cast<uintptr_t>(orow) >> 2) & 3) == (uintptr_t)r.mis) {  // the output's quads line up with the input's
      float4* oq = reinterpret_cast<float4*>(orow - r.mis);
      scan_quads(r, [&](int qi, int i0, const float (&x)[4]) {
        if (qi >= r.nq) return;
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = key_of(x[j]) >= thr ? x[j] * inv : 0.f;
        if (i0 >= 0 && i0 + 4 <= V) {
          oq[qi] = make_float4(y[0], y[1], y[2], y[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((unsigned)(i0 + j) < (unsigned)V) orow[i0 + j] = y[j];
        }
      });
    } else {
      scan_row(r, [&](int i, float x) { orow[i] = key_of(x) >= thr ? x * inv : 0.f; });
    }
    return;
  }
  // ---- the draw: the first element, in row order, at which the running kept mass exceeds target = floor(u * Z)
  const Philox rng(p.seed_ptr ? (uint64_t)*p.seed_ptr : p.seed, p.offset_ptr ? (uint64_t)*p.offset_ptr + p.offset : p.offset, (uint64_t)b);
  const unsigned long long u = ((unsigned long long)rng.c[0] << 32) | rng.c[1];
  const unsigned long long target = __umul64hi(u, Z);  // in [0, Z)
  int W = -1;  // the wave whose quads hold it (uniform: every thread reads the same 16 totals)
  unsigned long long wbefore = 0;
  {
    unsigned long long acc = slice_before;  // (cluster form: a slice that does not hold the target finds no wave and writes nothing)
    for (int i = 0; i < kT / 64; ++i) {
      const unsigned long long t = s.red64[i];
      if (W < 0 && t > 0 && acc <= target && target < acc + t) { W = i; wbefore = acc; }
      acc += t;
    }
  }
  if (threadIdx.x == 0) { s.out_idx = -1; s.pick_block = -1; }
  __syncthreads();
  if (W >= 0) {
    // that wave's quads in runs of G * 64 (G = 1 up to 4M elements): all waves add up run masses, one block scan finds the run,
    // one wave scans the run
    const int wq0 = r.qb + W * QW < r.nq ? r.qb + W * QW : r.nq, wq1 = wq0 + QW < r.nq ? wq0 + QW : r.nq;
    const int steps = (wq1 - wq0 + 63) >> 6, G = (steps + kBlocks - 1) / kBlocks, runs = (steps + G - 1) / G;
    for (int run = wave; run < runs; run += kT / 64) {
      unsigned long long m = 0;
      for (int g = 0; g < G; ++g) {
        const int qi = wq0 + (run * G + g) * 64 + lane;
        m += kept_mass(qi, wq1, r.q[qi < wq1 ? qi : 0]);
      }
      m = wave_sum_u64(m);
      if (lane == 0) s.bsum[run] = m;
    }
    __syncthreads();
    const unsigned long long mine = (int)threadIdx.x < runs ? s.bsum[threadIdx.x] : 0ull;
    unsigned long long Zw;
    const unsigned long long suf = block_suffix_sum<unsigned long long>(mine, s.red64, Zw);
    const unsigned long long before = wbefore + Zw - suf;  // mass before my run
    if (mine > 0 && before <= target && target < before + mine) {
      s.pick_block = (int)threadIdx.x;
      s.pick_before = before;
    }
    __syncthreads();
    if (wave == 0 && s.pick_block >= 0) {
      unsigned long long run_mass = s.pick_before;
      bool done = false;
      for (int g = 0; g < G && !done; ++g) {
        const int qi = wq0 + (s.pick_block * G + g) * 64 + lane;
        const float4 v = r.q[qi < wq1 ? qi : 0];
        const float x[4] = {v.x, v.y, v.z, v.w};
        const int i0 = qi * 4 - r.mis;
        unsigned long long m[4], m4 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          m[j] = (qi < wq1 && (unsigned)(i0 + j) < (unsigned)V && key_of(x[j]) >= thr) ? fix_of(x[j]) : 0ull;
          m4 += m[j];
        }
        unsigned long long incl = m4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned long long t = __shfl_up(incl, o, 64);
          if (lane >= o) incl += t;
        }
        const unsigned long long lo_mass = run_mass + incl - m4;
        const bool hit = m4 > 0 && lo_mass <= target && target < lo_mass + m4;
        if (hit) {
          unsigned long long acc = lo_mass;
          int pick = i0 + 3;
          bool found = false;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!found && m[j] > 0) {
              acc += m[j];
              if (acc > target) { pick = i0 + j; found = true; }
            }
          }
          s.out_idx = pick;
        }
        done = __any(hit);
        run_mass += __shfl(incl, 63, 64);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int r_ = s.out_idx;
    if constexpr (STAGE == 5) {
      if (r_ >= 0) p.out[b] = r_;  // (the one slice that holds the target)
      else if (Z == 0 && slice == 0) p.out[b] = 0;  // an all-zero row has no mass to draw from
    } else {
      if (r_ < 0) r_ = 0;  // an all-zero row has no mass to draw from
      p.out[b] = r_;
    }
  }
}

// Cluster form: up to 64 rows of a long vocabulary (a decode batch) leave most CUs idle with one workgroup per row, and a row pass
// is bound by that one CU's vector issue (63 - 113 us at 128k entries). With a scratch buffer the row is cut into `slices` runs,
// one workgroup each, and the passes become launches (see sampling_kernel). Returns the slice count (0: one workgroup per row).
static int cluster_slices(int64_t batch, int64_t vocab) {
  // (lease zu: vocab 128256 - 32 slices at 1 .. 8 rows 40 / 82 / 52 us for top-k renorm / top-k-first draw / joint draw against 63 / 113 / 94
  //  on one workgroup per row; 8 slices at 32 rows 48 / 95 / 60 against 68 / 121 / 96; 4 slices at 64 rows 62 / 118 / 72 against 73 / 127 / 99;
  //  2 slices at 96 and 128 rows lose. A launch costs ~4 us: six of them are most of what is left at one row.)
  if (batch <= 0 || batch > 64 || vocab < 32768) return 0;
  int64_t sl = num_cus() / batch;
  sl = sl > 32 ? 32 : sl;
  const int64_t by_len = (vocab / 4) / 512;  // at least 512 quads per slice
  sl = sl > by_len ? by_len : sl;
  return sl >= 2 ? (int)sl : 0;
}

static int run(hipStream_t st, const SampleParams& p0, int64_t batch, const char* op, void* ws = nullptr, int64_t ws_bytes = 0) {
  SGLK_REQUIRE(p0.V > 0, "%s: vocab size must be positive", op);
  if (batch == 0) return SGLK_OK;
  const int slices = ws != nullptr ? cluster_slices(batch, p0.V) : 0;
  if (slices > 1 && ws_bytes >= batch * (int64_t)sizeof(RowWs) && (uintptr_t)ws % 8 == 0) {
    SampleParams p = p0;
    p.ws = reinterpret_cast<RowWs*>(ws);
    p.slices = slices;
    if (hipMemsetAsync(ws, 0, (size_t)batch * sizeof(RowWs), st) != hipSuccess) return check_launch(op);
    const dim3 grid((unsigned)slices, (unsigned)batch);
    sampling_kernel<1><<<grid, kT, 0, st>>>(p);
    if (!p.use_minp) {
      sampling_kernel<2><<<grid, kT, 0, st>>>(p);
      sampling_kernel<3><<<grid, kT, 0, st>>>(p);
    }
    sampling_kernel<4><<<grid, kT, 0, st>>>(p);
    sampling_kernel<5><<<grid, kT, 0, st>>>(p);
    return check_launch(op);
  }
  sampling_kernel<0><<<(unsigned)batch, kT, 0, st>>>(p0);
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

extern "C" int sglk_top_k_top_p_sampling_from_probs_graph(sglk_stream_t stream, int32_t* output, const float* probs,
                                                          const int64_t* indices, const int32_t* top_k_arr, int64_t top_k_val,
                                                          const float* top_p_arr, float top_p_val, int use_top_k, int64_t batch,
                                                          int64_t vocab, const int64_t* philox_seed_ptr,
                                                          const int64_t* philox_offset_ptr, uint64_t offset_intragraph) {
  using namespace sglk;
  SGLK_REQUIRE(philox_seed_ptr && philox_offset_ptr, "top_k_top_p_sampling_from_probs_graph: the generator state pointers are NULL");
  SampleParams p{};
  p.probs = probs; p.out = output; p.indices = indices; p.k_arr = top_k_arr; p.k_is_i64 = 0; p.k_val = top_k_val;
  p.p_arr = top_p_arr; p.p_val = top_p_val; p.V = (int)vocab; p.use_k = use_top_k; p.use_p = 1; p.do_sample = 1;
  p.seed_ptr = philox_seed_ptr; p.offset_ptr = philox_offset_ptr; p.offset = offset_intragraph;
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

extern "C" int sglk_min_p_sampling_from_probs_graph(sglk_stream_t stream, int32_t* output, const float* probs, const int64_t* indices,
                                                    const float* min_p_arr, float min_p_val, int64_t batch, int64_t vocab,
                                                    const int64_t* philox_seed_ptr, const int64_t* philox_offset_ptr,
                                                    uint64_t offset_intragraph) {
  using namespace sglk;
  SGLK_REQUIRE(philox_seed_ptr && philox_offset_ptr, "min_p_sampling_from_probs_graph: the generator state pointers are NULL");
  SampleParams p{};
  p.probs = probs; p.out = output; p.indices = indices; p.p_arr = min_p_arr; p.p_val = min_p_val; p.V = (int)vocab;
  p.use_minp = 1; p.do_sample = 1; p.seed_ptr = philox_seed_ptr; p.offset_ptr = philox_offset_ptr; p.offset = offset_intragraph;
  return run((hipStream_t)stream, p, batch, "min_p_sampling_from_probs");
}


extern "C" int64_t sglk_sampling_workspace_size(int64_t batch, int64_t vocab) {
  using namespace sglk;
  return cluster_slices(batch, vocab) > 1 ? batch * (int64_t)sizeof(RowWs) : 0;
}

extern "C" int sglk_sampling_ws(sglk_stream_t stream, int op, void* result, const float* probs, const int64_t* indices,
                                const void* top_k_arr, int top_k_is_int64, int64_t top_k_val, const float* p_arr, float p_val,
                                int64_t batch, int64_t vocab, uint64_t philox_seed, uint64_t philox_offset,
                                const int64_t* philox_seed_ptr, const int64_t* philox_offset_ptr, void* workspace,
                                int64_t workspace_bytes) {
  using namespace sglk;
  SGLK_REQUIRE(op >= 0 && op <= 4, "sampling_ws: op must be 0 (top-k renorm), 1 (top-p renorm), 2 (top-k + top-p draw), 3 (top-p draw) or 4 (min-p draw)");
  SGLK_REQUIRE((philox_seed_ptr == nullptr) == (philox_offset_ptr == nullptr), "sampling_ws: the generator state pointers come as a pair");
  SampleParams p{};
  p.probs = probs; p.V = (int)vocab; p.indices = indices;
  p.k_arr = top_k_arr; p.k_is_i64 = top_k_is_int64; p.k_val = top_k_val; p.p_arr = p_arr; p.p_val = p_val;
  p.seed = philox_seed; p.offset = philox_offset; p.seed_ptr = philox_seed_ptr; p.offset_ptr = philox_offset_ptr;
  const char* name = "sampling";
  switch (op) {
    case 0: p.renorm = (float*)result; p.use_k = 1; p.indices = nullptr; name = "top_k_renorm_probs"; break;
    case 1: p.renorm = (float*)result; p.use_p = 1; p.indices = nullptr; name = "top_p_renorm_probs"; break;
    case 2: p.out = (int32_t*)result; p.use_k = 1; p.use_p = 1; p.do_sample = 1; name = "top_k_top_p_sampling_from_probs"; break;
    case 3: p.out = (int32_t*)result; p.use_p = 1; p.do_sample = 1; name = "top_p_sampling_from_probs"; break;
    default: p.out = (int32_t*)result; p.use_minp = 1; p.do_sample = 1; name = "min_p_sampling_from_probs"; break;
  }
  return run((hipStream_t)stream, p, batch, name, workspace, workspace_bytes);
}
