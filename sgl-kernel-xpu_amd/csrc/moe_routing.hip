// MoE routing and data movement for gfx950: topk_softmax, moe_align_block_size,
// prepare_moe_input, scatter_tokens_to_experts, apply_shuffle_mul_sum.
//
// Replaces reference src/sycl/TopKSoftMax.cpp:584-644, src/sycl/MoEAlign.cpp:313-383 and
// src/sycl/MoEPrepareInputs.cpp:459-497, :571-589, :691-755. Semantics kept:
//   topk_softmax      softmax over experts in fp32; k rounds of arg-max with strict '>' so the lower
//                     index wins ties (TopKSoftMax.cpp:395-421); weights are the selected probabilities,
//                     optionally times 1/sum (:462-470). The arg-max runs on the LOGITS, which orders
//                     experts exactly like their probabilities (exp is monotone) and, unlike the
//                     probabilities, cannot collapse distinct experts into a tie by underflow or
//                     rounding -- so the index path is exact and independent of the exp implementation.
//   moe_align         bucket = id + 1 (id -1 = "not on this rank" lands in bucket 0); per-bucket
//                     counts padded to block_size, exclusive scan -> cumsum[0..num_experts];
//                     expert_ids[block] = bucket - 1; sorted_token_ids filled by atomic ranks
//                     (order inside a bucket is unspecified, as in the reference :35-58); cumsum is left
//                     holding prefix + count like the reference's second kernel leaves it.
//   prepare_moe_input expert_offsets[e] = ROW COUNT of expert e (not an offset: MoEPrepareInputs.cpp:52),
//                     problem_sizes1[e] = (count, 2n, k), problem_sizes2[e] = (count, k, n),
//                     input_permutation[dst] = src token, output_permutation[slot] = dst row.
//                     Here the order inside an expert is the flat slot order (deterministic; the
//                     reference's atomics make it run-dependent).
//   scatter / combine row copies and fp32 weighted sums (acc += float(x) * w [* rsf]) in slot order.
// These are latency / HBM-copy kernels; the design rule is "one launch, no temporaries, 16-byte rows".
#include <math.h>

#include "common.h"

namespace sglk {
namespace {

// ---------------------------------------------------------------------------------- topk_softmax
// One wave per token; lane l owns experts l, l+64, l+128, l+192 (E <= 256).
template <typename T>
__global__ __launch_bounds__(256) void topk_softmax_kernel(float* __restrict__ weights, int* __restrict__ indices,
                                                           const T* __restrict__ gating, int64_t tokens, int E,
                                                           int k, bool renorm) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= tokens) return;
  const T* row = gating + tok * E;
  float x[4], v[4];  // logits (selection key) and probabilities (weights)
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = lane + 64 * i;
    x[i] = e < E ? (float)row[e] : -INFINITY;
    mx = fmaxf(mx, x[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (lane + 64 * i < E) ? expf(x[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] *= inv;

  float picked_sum = 0.f;
  float my_w = 0.f;  // lane j < k keeps the j-th selected weight
  int my_i = 0;
  for (int j = 0; j < k; ++j) {
    // local arg-max: ascending expert index with strict '>' keeps the lower index on ties
    float bx = x[0], bv = v[0];
    int be = lane;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (x[i] > bx) { bx = x[i]; bv = v[i]; be = lane + 64 * i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ox = __shfl_xor(bx, o, 64);
      const float ov = __shfl_xor(bv, o, 64);
      const int oe = __shfl_xor(be, o, 64);
      if (ox > bx || (ox == bx && oe < be)) { bx = ox; bv = ov; be = oe; }
    }
    if (lane == j) { my_w = bv; my_i = be; }
    picked_sum += bv;
    if ((be & 63) == lane) x[be >> 6] = -INFINITY;  // clear the winner
  }
  if (lane < k) {
    const float w = renorm ? my_w * (1.0f / picked_sum) : my_w;
    weights[tok * k + lane] = w;
    indices[tok * k + lane] = my_i;
  }
}

// ------------------------------------------------------------------------------- moe_align_block_size
// Counting of a prefill-sized id list by many workgroups (one workgroup takes ~1 us per 1024 ids: 131072 ids - 16384 tokens, top-8 -
// were 90 - 127 us): LDS counters per workgroup, flushed into `cumsum` (zeroed by the caller), which moe_align_kernel<PRE> then reads.
template <typename IdT>
__global__ __launch_bounds__(1024) void moe_align_count_kernel(const IdT* __restrict__ topk_ids, int32_t* cumsum, int num_experts,
                                                               int64_t numel) {
  extern __shared__ int32_t sh[];
  const int tid = threadIdx.x;
  for (int i = tid; i < num_experts; i += 1024) sh[i] = 0;
  __syncthreads();
  const bool aggregate = num_experts <= 32;
  for (int64_t i0 = (int64_t)blockIdx.x * 8 * 1024; i0 < numel; i0 += (int64_t)gridDim.x * 8 * 1024) {
    int bs[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t i = i0 + r * 1024 + tid;
      const int b = i < numel ? (int)topk_ids[i] + 1 : -1;
      bs[r] = b < num_experts ? b : -1;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int b = bs[r];
      if (!aggregate) {
        if (b >= 0) atomicAdd(&sh[b], 1);
        continue;
      }
      unsigned long long todo = __ballot(b >= 0);
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int b0 = __shfl(b, leader, 64);
        const unsigned long long same = __ballot(b == b0);
        if ((tid & 63) == leader) atomicAdd(&sh[b0], __builtin_popcountll(same));
        todo &= ~same;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < num_experts; i += 1024)
    if (sh[i]) atomicAdd(&cumsum[i], sh[i]);
}

template <typename IdT, bool PRE = false>  // PRE: the bucket counts are in cumsum[0 .. num_experts) (moe_align_count_kernel)
__global__ __launch_bounds__(1024) void moe_align_kernel(const IdT* __restrict__ topk_ids, int32_t* sorted_token_ids,
                                                         int32_t* expert_ids, int32_t* total_tokens_post_pad,
                                                         int32_t* cumsum, int num_experts, int block_size,
                                                         int64_t numel, bool pad_sorted) {
  extern __shared__ int32_t sh[];  // counts[num_experts], prefix[num_experts + 1]
  int32_t* counts = sh;
  int32_t* prefix = sh + num_experts;
  const int tid = threadIdx.x;
  for (int i = tid; i < num_experts; i += 1024) counts[i] = PRE ? cumsum[i] : 0;
  __syncthreads();

  // Counting. Eight ids per thread are requested before any is used (one memory round trip per 8192 ids instead of one per
  // 1024: the single workgroup is latency-bound). With few buckets (a top-2-of-8 routing puts 8192 ids on 9 counters)
  // the lanes of a wave that hit the same bucket send ONE atomic; with many buckets plain atomics collide rarely and the
  // aggregation loop (one trip per distinct bucket of the wave) would cost more than it saves.
  const bool aggregate = num_experts <= 32;
  for (int64_t i0 = 0; i0 < (PRE ? 0 : numel); i0 += 8 * 1024) {
    int bs[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t i = i0 + r * 1024 + tid;
      const int b = i < numel ? (int)topk_ids[i] + 1 : -1;
      bs[r] = b < num_experts ? b : -1;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int b = bs[r];
      if (!aggregate) {
        if (b >= 0) atomicAdd(&counts[b], 1);
        continue;
      }
      unsigned long long todo = __ballot(b >= 0);
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int b0 = __shfl(b, leader, 64);
        const unsigned long long same = __ballot(b == b0);
        if ((tid & 63) == leader) atomicAdd(&counts[b0], __builtin_popcountll(same));
        todo &= ~same;
      }
    }
  }
  __syncthreads();
  // exclusive scan of the padded counts by one wave (num_experts is at most a few hundred)
  if (tid < 64) {
    int carry = 0;
    for (int base = 0; base < num_experts; base += 64) {
      const int i = base + tid;
      const int c = i < num_experts ? counts[i] : 0;
      const int padded = (c + block_size - 1) / block_size * block_size;
      int incl = padded;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (tid >= o) incl += t;
      }
      if (i < num_experts) prefix[i] = carry + incl - padded;
      carry += __shfl(incl, 63, 64);
    }
    if (tid == 0) {
      prefix[num_experts] = carry;
      *total_tokens_post_pad = carry;
    }
  }
  __syncthreads();
  const int total = prefix[num_experts];
  for (int i = tid; i <= num_experts; i += 1024) cumsum[i] = prefix[i];
  const int num_blocks = total / block_size;
  for (int i = tid; i < num_blocks; i += 1024) {
    const int start = i * block_size;
    int left = 0, right = num_experts;
    while (left < right) {
      const int mid = (left + right) >> 1;
      if (prefix[mid] <= start) left = mid + 1; else right = mid;
    }
    expert_ids[i] = left - 2;
  }
  if (pad_sorted) {
    for (int i = tid; i < total; i += 1024) sorted_token_ids[i] = (int32_t)numel;
  }
}

template <typename IdT>
__global__ __launch_bounds__(256) void moe_align_sort_kernel(const IdT* __restrict__ topk_ids,
                                                             int32_t* sorted_token_ids, int32_t* cumsum,
                                                             int num_experts, int64_t numel) {
  // wave-aggregated ranks: one returning atomic per (wave, bucket) hands out a run of slots, the lanes take theirs by
  // their position among the wave's lanes of that bucket (the order inside a bucket is unspecified, as in the reference)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < numel; i0 += stride) {
    const int64_t i = i0 + threadIdx.x;
    int b = i < numel ? (int)topk_ids[i] + 1 : -1;
    if (b >= num_experts) b = -1;
    if (num_experts > 32) {  // many buckets: plain returning atomics (see the counting kernel)
      if (b >= 0) sorted_token_ids[atomicAdd(&cumsum[b], 1)] = (int32_t)i;
      continue;
    }
    unsigned long long todo = __ballot(b >= 0);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const int b0 = __shfl(b, leader, 64);
      const unsigned long long same = __ballot(b == b0);
      int base = 0;
      if (lane == leader) base = atomicAdd(&cumsum[b0], __builtin_popcountll(same));
      base = __shfl(base, leader, 64);
      if (b == b0) sorted_token_ids[base + __builtin_popcountll(same & ((1ull << lane) - 1ull))] = (int32_t)i;
      todo &= ~same;
    }
  }
}

// ---------------------------------------------------------------------------------- prepare_moe_input
// One 256-thread block per expert: pass 1 counts ids == e and ids < e (= this expert's first row),
// pass 2 hands out stable ranks in flat slot order. No atomics, no temporaries, deterministic.
template <typename IdT>
__global__ __launch_bounds__(256) void prepare_moe_input_kernel(const IdT* __restrict__ topk_ids,
                                                                IdT* expert_counts, IdT* problem_sizes1,
                                                                IdT* problem_sizes2, IdT* input_perm,
                                                                IdT* output_perm, int64_t numel, int topk, int n,
                                                                int k) {
  __shared__ int red[2][4];
  __shared__ int wave_tot[4];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int eq = 0, lt = 0;
  for (int64_t i = tid; i < numel; i += 256) {
    const int id = (int)topk_ids[i];
    eq += (id == e);
    lt += (id >= 0 && id < e);
  }
  eq = (int)wave_sum((float)eq);  // counts < 2^24: exact in fp32
  lt = (int)wave_sum((float)lt);
  if (lane == 0) { red[0][wave] = eq; red[1][wave] = lt; }
  __syncthreads();
  const int count = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const int first = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  if (tid == 0) {
    expert_counts[e] = (IdT)count;
    problem_sizes1[e * 3 + 0] = (IdT)count;
    problem_sizes1[e * 3 + 1] = (IdT)(2 * n);
    problem_sizes1[e * 3 + 2] = (IdT)k;
    problem_sizes2[e * 3 + 0] = (IdT)count;
    problem_sizes2[e * 3 + 1] = (IdT)k;
    problem_sizes2[e * 3 + 2] = (IdT)n;
  }
  int base = first;
  for (int64_t i0 = 0; i0 < numel; i0 += 256) {
    const int64_t i = i0 + tid;
    const bool hit = i < numel && (int)topk_ids[i] == e;
    const unsigned long long m = __ballot(hit);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (hit) {
      const int pos = base + before + in_wave;
      input_perm[pos] = (IdT)(i / topk);
      output_perm[i] = (IdT)pos;
    }
    base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  }
}

// ---------------------------------------------------------------------------- scatter_tokens_to_experts
// grid = tokens; each thread moves 16-byte pieces of the source row to its topk destination rows.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                           const int32_t* __restrict__ src2dst, int topk,
                                                           int row_vecs) {
  const int64_t tok = blockIdx.x;
  for (int c = threadIdx.x; c < row_vecs; c += 256) {
    const uint4 v = in[tok * row_vecs + c];
    for (int j = 0; j < topk; ++j) out[(int64_t)src2dst[tok * topk + j] * row_vecs + c] = v;
  }
}

// -------------------------------------------------------------------------------- apply_shuffle_mul_sum
template <typename T, typename W>
__global__ __launch_bounds__(256) void shuffle_mul_sum_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                              const int32_t* __restrict__ perm,
                                                              const W* __restrict__ factors, int topk, int hidden,
                                                              float rsf, bool use_rsf) {
  constexpr int V = 16 / sizeof(T);
  const int64_t tok = blockIdx.x;
  // (a workgroup per token and 256 * V columns, the rows of up to eight top-k slots requested before the first is used: one token of
  //  a top-8 router over 7168 columns was 32 dependent round trips - 13 us; the sum keeps its slot order)
  const int c = ((int)blockIdx.y * 256 + (int)threadIdx.x) * V;
  if (c >= hidden) return;
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  for (int j0 = 0; j0 < topk; j0 += 8) {
    Vec<T, V> x[8];
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u < topk ? j0 + u : topk - 1;
      const int64_t src = perm[tok * topk + j];
      w[u] = factors ? (float)factors[tok * topk + j] : 1.0f;
      x[u] = load_vec<T, V>(in + src * hidden + c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (j0 + u < topk) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          // explicit roundings (no fma contraction): acc += (x * w) [* rsf], as the reference spells it
          float t = __fmul_rn((float)x[u][i], w[u]);
          if (use_rsf) t = __fmul_rn(t, rsf);
          acc[i] = __fadd_rn(acc[i], t);
        }
      }
    }
  }
  Vec<T, V> o;
#pragma unroll
  for (int i = 0; i < V; ++i) o[i] = (T)acc[i];
  store_vec<T, V>(out + tok * hidden + c, o);
}

// The same combine behind a down projection whose K range was split in two (moe_persist.hip, KSPL = 2): source row r of a full
// row block (block = 128 or 256 rows) is T(ws[0][r] + ws[1][r]) - the one rounding the GEMM's own store would have made - and a
// row of an expert's remainder of 1 .. block / 2 rows (the streaming kernels' share) is y[r]. Which one: from rows_per_expert, as
// the GEMM's tile walk does.
template <typename T, typename W>
__global__ __launch_bounds__(256) void shuffle_mul_sum_splitk_kernel(const T* __restrict__ y, const float* __restrict__ ws,
                                                                     T* __restrict__ out, const int32_t* __restrict__ perm,
                                                                     const W* __restrict__ factors,
                                                                     const int32_t* __restrict__ rows_per_expert, int E,
                                                                     int block, int64_t total_m, int topk, int hidden,
                                                                     float rsf, bool use_rsf) {
  constexpr int V = 4;  // 16 bytes of fp32 per lane and slab
  const int64_t tok = blockIdx.x;
  __shared__ int s_tail[16];  // per top-k slot: 1 = the row is in y
  if ((int)threadIdx.x < topk) {
    const int src = perm[tok * topk + threadIdx.x];
    int base = 0, tail = 0;
    for (int e = 0; e < E; ++e) {
      const int r = rows_per_expert[e];
      if (src >= base && src < base + r) {
        const int rem = r & (block - 1);
        tail = (rem >= 1 && rem <= block / 2 && src - base >= r - rem) ? 1 : 0;
        break;
      }
      base += r;
    }
    s_tail[threadIdx.x] = tail;
  }
  __syncthreads();
  for (int c = threadIdx.x * V; c < hidden; c += 256 * V) {
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    for (int j = 0; j < topk; ++j) {
      const int64_t src = perm[tok * topk + j];
      const float w = factors ? (float)factors[tok * topk + j] : 1.0f;
      Vec<T, V> x;
      if (s_tail[j]) {
        x = load_vec<T, V>(y + src * hidden + c);
      } else {
        const Vec<float, V> p0 = load_vec<float, V>(ws + src * hidden + c);
        const Vec<float, V> p1 = load_vec<float, V>(ws + (total_m + src) * hidden + c);
#pragma unroll
        for (int i = 0; i < V; ++i) x[i] = (T)__fadd_rn(p0[i], p1[i]);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float t = __fmul_rn((float)x[i], w);
        if (use_rsf) t = __fmul_rn(t, rsf);
        acc[i] = __fadd_rn(acc[i], t);
      }
    }
    Vec<T, V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = (T)acc[i];
    store_vec<T, V>(out + tok * hidden + c, o);
  }
}

// Prefill-sized lists: a workgroup ranks its 8192 ids per bucket with returning LDS atomics and reserves one run per bucket with
// ONE global atomic (131072 ids on 128 buckets were 131072 returning global atomics on 128 addresses - most of the op's time).
template <typename IdT>
__global__ __launch_bounds__(1024) void moe_align_sort_chunk_kernel(const IdT* __restrict__ topk_ids, int32_t* sorted_token_ids,
                                                                    int32_t* cumsum, int num_experts, int64_t numel) {
  extern __shared__ int32_t sh[];  // cnt[num_experts], base[num_experts]
  int32_t* cnt = sh;
  int32_t* base = sh + num_experts;
  const int tid = threadIdx.x;
  for (int i = tid; i < num_experts; i += 1024) cnt[i] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * 8 * 1024;
  int bs[8], rk[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int64_t i = i0 + r * 1024 + tid;
    const int b = i < numel ? (int)topk_ids[i] + 1 : -1;
    bs[r] = b < num_experts ? b : -1;
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) rk[r] = bs[r] >= 0 ? atomicAdd(&cnt[bs[r]], 1) : 0;
  __syncthreads();
  for (int i = tid; i < num_experts; i += 1024) base[i] = cnt[i] ? atomicAdd(&cumsum[i], cnt[i]) : 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (bs[r] >= 0) sorted_token_ids[base[bs[r]] + rk[r]] = (int32_t)(i0 + r * 1024 + tid);
}

template <typename IdT>
static int align_launch(hipStream_t st, const void* ids, int32_t* sorted, int32_t* eids, int32_t* total,
                        int32_t* cumsum, int num_experts, int block_size, int64_t numel, bool pad) {
  const size_t lds = (size_t)(2 * num_experts + 1) * sizeof(int32_t);
  if (numel >= 16384) {  // prefill-sized: count on many CUs (three more launch slots, ~100 us less at 131072 ids)
    if (hipMemsetAsync(cumsum, 0, (size_t)num_experts * sizeof(int32_t), st) != hipSuccess) return check_launch("moe_align_block_size(zero)");
    const int64_t wgs = cdiv(numel, 8 * 1024);
    moe_align_count_kernel<IdT><<<(unsigned)(wgs < 256 ? wgs : 256), 1024, (size_t)num_experts * sizeof(int32_t), st>>>(
        (const IdT*)ids, cumsum, num_experts, numel);
    if (int rc = check_launch("moe_align_block_size(count)")) return rc;
    moe_align_kernel<IdT, true><<<1, 1024, lds, st>>>((const IdT*)ids, sorted, eids, total, cumsum, num_experts, block_size, numel, pad);
  } else {
    moe_align_kernel<IdT><<<1, 1024, lds, st>>>((const IdT*)ids, sorted, eids, total, cumsum, num_experts, block_size, numel, pad);
  }
  if (int rc = check_launch("moe_align_block_size")) return rc;
  if (numel >= 16384 && cdiv(numel, 8 * 1024) < (1ll << 31)) {
    moe_align_sort_chunk_kernel<IdT><<<(unsigned)cdiv(numel, 8 * 1024), 1024, (size_t)2 * num_experts * sizeof(int32_t), st>>>(
        (const IdT*)ids, sorted, cumsum, num_experts, numel);
    return check_launch("moe_align_block_size(sort)");
  }
  if (numel > 0) {
    const int64_t want = cdiv(numel, 256);
    moe_align_sort_kernel<IdT><<<(unsigned)(want < 1024 ? want : 1024), 256, 0, st>>>((const IdT*)ids, sorted, cumsum,
                                                                                      num_experts, numel);
    return check_launch("moe_align_block_size(sort)");
  }
  return SGLK_OK;
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_topk_softmax(sglk_stream_t stream, float* topk_weights, int32_t* topk_indices,
                                 const void* gating, int64_t tokens, int64_t experts, int64_t topk, int renormalize,
                                 int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(experts > 0 && experts <= 256, "n_experts only support up to 256, but got %lld", (long long)experts);
  SGLK_REQUIRE(topk > 0 && topk <= experts && topk <= 64, "n_topk must satisfy 0 < n_topk <= min(n_experts, 64)");
  if (tokens == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)cdiv(tokens, 4);
  SGLK_DISPATCH_FLOAT(dtype, T, {
    topk_softmax_kernel<T><<<blocks, 256, 0, st>>>(topk_weights, topk_indices, (const T*)gating, tokens,
                                                   (int)experts, (int)topk, renormalize != 0);
  });
  return check_launch("topk_softmax");
}

extern "C" int sglk_moe_align_block_size(sglk_stream_t stream, const void* topk_ids, int ids_dtype, int64_t numel,
                                         int64_t num_experts, int64_t block_size, int32_t* sorted_token_ids,
                                         int32_t* expert_ids, int32_t* num_tokens_post_pad, int32_t* cumsum_buffer,
                                         int pad_sorted_token_ids) {
  using namespace sglk;
  SGLK_REQUIRE(num_experts > 0 && num_experts <= 8000, "moe_align_block_size: num_experts must be in [1, 8000]");
  SGLK_REQUIRE(block_size > 0, "moe_align_block_size: block_size must be positive");
  hipStream_t st = (hipStream_t)stream;
  if (ids_dtype == SGLK_I32)
    return align_launch<int32_t>(st, topk_ids, sorted_token_ids, expert_ids, num_tokens_post_pad, cumsum_buffer,
                                 (int)num_experts, (int)block_size, numel, pad_sorted_token_ids != 0);
  if (ids_dtype == SGLK_I64)
    return align_launch<int64_t>(st, topk_ids, sorted_token_ids, expert_ids, num_tokens_post_pad, cumsum_buffer,
                                 (int)num_experts, (int)block_size, numel, pad_sorted_token_ids != 0);
  return fail(SGLK_EUNSUPPORTED, "moe_align_block_size: topk_ids must be int32 or int64");
}

extern "C" int sglk_prepare_moe_input(sglk_stream_t stream, const void* topk_ids, void* expert_counts,
                                      void* problem_sizes1, void* problem_sizes2, void* input_permutation,
                                      void* output_permutation, int64_t numel, int64_t topk, int64_t num_experts,
                                      int64_t n, int64_t k, int ids_dtype) {
  using namespace sglk;
  SGLK_REQUIRE(num_experts > 0 && topk > 0, "prepare_moe_input: num_experts and topk must be positive");
  hipStream_t st = (hipStream_t)stream;
  if (ids_dtype == SGLK_I32) {
    prepare_moe_input_kernel<int32_t><<<(unsigned)num_experts, 256, 0, st>>>(
        (const int32_t*)topk_ids, (int32_t*)expert_counts, (int32_t*)problem_sizes1, (int32_t*)problem_sizes2,
        (int32_t*)input_permutation, (int32_t*)output_permutation, numel, (int)topk, (int)n, (int)k);
  } else if (ids_dtype == SGLK_I64) {
    prepare_moe_input_kernel<int64_t><<<(unsigned)num_experts, 256, 0, st>>>(
        (const int64_t*)topk_ids, (int64_t*)expert_counts, (int64_t*)problem_sizes1, (int64_t*)problem_sizes2,
        (int64_t*)input_permutation, (int64_t*)output_permutation, numel, (int)topk, (int)n, (int)k);
  } else {
    return fail(SGLK_EUNSUPPORTED, "prepare_moe_input: index tensors must be int32 or int64");
  }
  return check_launch("prepare_moe_input");
}

extern "C" int sglk_scatter_tokens_to_experts(sglk_stream_t stream, const void* input, const int32_t* src2dst_map,
                                              void* output, int64_t tokens, int64_t topk, int64_t row_bytes) {
  using namespace sglk;
  SGLK_REQUIRE(row_bytes % 16 == 0 && (uintptr_t)input % 16 == 0 && (uintptr_t)output % 16 == 0,
               "scatter_tokens_to_experts: rows must be multiples of 16 bytes");
  SGLK_REQUIRE(topk > 0 && topk <= 16, "scatter_tokens_to_experts: topk must be in [1, 16]");
  if (tokens == 0) return SGLK_OK;
  scatter_rows_kernel<<<(unsigned)tokens, 256, 0, (hipStream_t)stream>>>((const uint4*)input, (uint4*)output,
                                                                         src2dst_map, (int)topk, (int)(row_bytes / 16));
  return check_launch("scatter_tokens_to_experts");
}

extern "C" int sglk_apply_shuffle_mul_sum(sglk_stream_t stream, const void* input, void* output,
                                          const int32_t* permutation, const void* factors, int64_t tokens,
                                          int64_t topk, int64_t hidden, float routed_scaling_factor, int dtype,
                                          int factors_dtype) {
  using namespace sglk;
  SGLK_REQUIRE(topk > 0 && topk <= 16, "apply_shuffle_mul_sum: topk must be in [1, 16]");
  SGLK_REQUIRE(hidden > 0, "apply_shuffle_mul_sum: hidden size must be positive");
  if (tokens == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool use_rsf = routed_scaling_factor != 1.0f;
  SGLK_DISPATCH_FLOAT(dtype, T, {
    SGLK_REQUIRE((hidden * sizeof(T)) % 16 == 0 && (uintptr_t)input % 16 == 0 && (uintptr_t)output % 16 == 0,
                 "apply_shuffle_mul_sum: rows must be multiples of 16 bytes");
    if (factors == nullptr) {
      shuffle_mul_sum_kernel<T, T><<<dim3((unsigned)tokens, (unsigned)cdiv(hidden, 256 * (16 / (int64_t)sizeof(T)))), 256, 0, st>>>((const T*)input, (T*)output, permutation, nullptr,
                                                                     (int)topk, (int)hidden, routed_scaling_factor,
                                                                     use_rsf);
    } else {
      SGLK_DISPATCH_FLOAT(factors_dtype, W, {
        shuffle_mul_sum_kernel<T, W><<<dim3((unsigned)tokens, (unsigned)cdiv(hidden, 256 * (16 / (int64_t)sizeof(T)))), 256, 0, st>>>((const T*)input, (T*)output, permutation,
                                                                       (const W*)factors, (int)topk, (int)hidden,
                                                                       routed_scaling_factor, use_rsf);
      });
    }
  });
  return check_launch("apply_shuffle_mul_sum");
}

extern "C" int sglk_apply_shuffle_mul_sum_splitk(sglk_stream_t stream, const void* y, const float* ws, void* output,
                                                 const int32_t* permutation, const void* factors,
                                                 const int32_t* rows_per_expert, int64_t n_experts, int64_t block_rows,
                                                 int64_t total_m, int64_t tokens, int64_t topk, int64_t hidden,
                                                 float routed_scaling_factor, int dtype, int factors_dtype) {
  using namespace sglk;
  SGLK_REQUIRE(block_rows == 128 || block_rows == 256, "apply_shuffle_mul_sum_splitk: block_rows must be 128 or 256");
  SGLK_REQUIRE(topk > 0 && topk <= 16, "apply_shuffle_mul_sum_splitk: topk must be in [1, 16]");
  SGLK_REQUIRE(hidden > 0 && hidden % 8 == 0, "apply_shuffle_mul_sum_splitk: hidden size must be a positive multiple of 8");
  SGLK_REQUIRE(n_experts > 0 && n_experts < (1ll << 20) && total_m >= 0 && total_m < (1ll << 31),
               "apply_shuffle_mul_sum_splitk: bad n_experts / total_m");
  SGLK_REQUIRE((uintptr_t)y % 16 == 0 && (uintptr_t)ws % 16 == 0 && (uintptr_t)output % 16 == 0,
               "apply_shuffle_mul_sum_splitk: y, ws and output must be 16-byte aligned");
  if (tokens == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool use_rsf = routed_scaling_factor != 1.0f;
  SGLK_DISPATCH_HALF(dtype, T, {
    if (factors == nullptr) {
      shuffle_mul_sum_splitk_kernel<T, T><<<(unsigned)tokens, 256, 0, st>>>(
          (const T*)y, ws, (T*)output, permutation, nullptr, rows_per_expert, (int)n_experts, (int)block_rows, total_m, (int)topk,
          (int)hidden, routed_scaling_factor, use_rsf);
    } else {
      SGLK_DISPATCH_FLOAT(factors_dtype, W, {
        shuffle_mul_sum_splitk_kernel<T, W><<<(unsigned)tokens, 256, 0, st>>>(
            (const T*)y, ws, (T*)output, permutation, (const W*)factors, rows_per_expert, (int)n_experts, (int)block_rows, total_m,
            (int)topk, (int)hidden, routed_scaling_factor, use_rsf);
      });
    }
  });
  return check_launch("apply_shuffle_mul_sum_splitk");
}
