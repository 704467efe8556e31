// Flash-attention forward (`fwd` / mha_fwd) for gfx950: varlen Q, KV either ragged-contiguous or
// paged, GQA, causal (bottom-right aligned) / sliding window, softcap, sinks, split-KV decode, LSE.
//
// Replaces reference src/sycl/flash_attention.cpp:1332-1435 (router), :273-676 (paged decode),
// :83-270 / :686-876 (non-paged), :879-1214 (prefill), :1229-1328 (chunk-prefill) and the CuTe kernels in
// src/sycl/kernels/flash_attention_v2/. One kernel covers all of those routes: the reference needs separate
// decode / prefill / two-launch chunk-prefill kernels because its tiles are specialised per route; here the
// query rows of a (batch, kv-head) pair are "packed GQA" rows r = (q_pos, g) so decode (seqlen_q = 1) is just
// a short row list, and mixed batches run in a single launch without any host-side classification (the
// reference's reason for the two-launch scheme, flash_attention.cpp:1426-1429).
// Semantics (pinned by tests/test_flash_attention.py:349-476 attention_ref):
//   s = q.k * scale; softcap: s = cap * tanh(s / cap); masked if k >= seqlen_k, k > q_abs + right,
//   or (left >= 0 and k < q_abs - left) with q_abs = q_pos + seqlen_k - seqlen_q; optional per-head sink logit
//   joins the softmax denominator; rows with no visible key give 0.
//
// Design: workgroup = 4 waves, wave w owns 16 packed rows; 32-token K and V tiles are staged
// global -> registers -> LDS (issue early / write late, 3-slot LDS ring, one barrier per tile) so that
// any head dim (multiple of 8), page size, stride and mask shape goes through one path; S^T = K.Q^T with
// 16x16x32 MFMA keeps the softmax lane-local and P directly usable as the A operand of O += P.V; V is read
// with the hardware transpose read. LDS images are [256-byte column block][32 rows][256 B] with the chunk
// swizzle / k-permutation / token-permutation of mla_decode.hip (conflict-free row and transposed reads).
#include <math.h>

#include <type_traits>

#include "common.h"

namespace sglk {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

constexpr int kTile = 32;       // kv tokens per tile
constexpr int kRowsPerWave = 16;
constexpr int kWaves = 4;
constexpr int kBlockM = kRowsPerWave * kWaves;

#define SGLK_LDS(p) ((__attribute__((address_space(3))) void*)(p))

template <typename T>
struct Mfma;
template <>
struct Mfma<bf16> {
  static __device__ __forceinline__ v4f run(const v8s& a, const v8s& b, const v4f& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float x) { return __builtin_bit_cast(short, (bf16)x); }
};
template <>
struct Mfma<f16> {
  static __device__ __forceinline__ v4f run(const v8s& a, const v8s& b, const v4f& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float x) { return __builtin_bit_cast(short, (f16)x); }
};

typedef float v2f __attribute__((ext_vector_type(2)));

// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight (__syncthreads() also
// drains vmcnt, which would end the prefetch of the next tiles at every barrier)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int sw_main(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

struct AttnParams {
  void* out;            // [total_q, Hq, D]
  float* lse;           // [Hq, total_q]
  float* part_o;        // [splits, total_q, Hq, D] fp32 (splits > 1)
  float* part_lse;      // [splits, Hq, total_q]
  const float* sinks;   // [Hq] or null
  const float* k_descale;  // fp8 KV cache: one float for the whole K cache (device pointer) or null
  const float* v_descale;
  int64_t q_s0, q_s1;   // q strides (token, head) in elements
  int64_t o_s0, o_s1;
  int64_t k_s0, k_s1, k_s2;  // paged: (page, token-in-page, head); ragged: (token, head, -)
  int64_t v_s0, v_s1, v_s2;
  int64_t table_stride;
  int Hq, Hk, G, D, total_q;
  int page_shift;       // log2(page size), paged only
  int paged;            // 0: ragged k/v [total_k, Hk, D] + cu_seqlens_k; 1: paged [pages, page, Hk, D] + page_table + key END
                        // positions; 2: one cache row per slot [slots, seqlen_cache, Hk, D] + key END positions
  const int32_t* kv_batch_idx;  // [b] cache row (page-table row / slot) of sequence b, or null: b itself (layouts 1, 2)
  const int32_t* leftpad_k;     // [b] first valid cache position of sequence b, or null: 0 (layouts 1, 2)
  int causal_right;     // window right (>= 0 active, < 0 unlimited)
  int window_left;      // >= 0 active, < 0 unlimited
  int splits;
  int row_groups;       // decode kernel: 16-row groups of packed rows per (sequence, kv head), one workgroup each (1 .. 4)
  float scale;          // softmax scale
  float softcap;        // 0 = off
  int probe;            // libsglk_probes.so only (0 in the release library): attn_prefill_kernel timing probes, garbage results
  unsigned long long* stamps;  // libsglk_probes.so only: per-wave cycle sums of the prefill tile loop's segments, or null
};

// DKP: head dim rounded up to 32 (k-steps of the QK product); the V/O side uses ceil(D/16) 16-wide tiles.
// KV8: 0 = K/V stored like q (16-bit); 1 = fp8 e4m3fn, 2 = fp8 e5m2 cache (one byte per element, strides in bytes)
// converted to T while it is staged into LDS, with the per-tensor descale folded into the softmax scale (K) and the
// output normaliser (V) - reference flash_attention.cpp:561-572, tests/test_flash_attention.py:1697-1830.
template <typename T, int DKP, int KV8>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p, const T* __restrict__ q,
                                                       const char* __restrict__ kcache, const char* __restrict__ vcache,
                                                       const int32_t* __restrict__ cu_q,
                                                       const int32_t* __restrict__ seq_k,  // paged: lengths [b]; ragged: cu [b+1]
                                                       const int32_t* __restrict__ page_table) {
  using M = Mfma<T>;
  constexpr int KS = DKP / 32;                 // k-steps
  constexpr int NT = DKP / 16;                 // upper bound of 16-wide output tiles
  constexpr int NB = (DKP * 2 + 255) / 256;    // 256-byte column blocks per row
  constexpr int TILE_BYTES = NB * kTile * 256; // one operand, one slot
  constexpr int SLOT = 2 * TILE_BYTES;         // K then V
  constexpr int kSlots = DKP > 256 ? 2 : 3;    // LDS ring depth (d = 512 tiles are 64 KiB: two slots, two barriers)
  constexpr int CPR_MAX = DKP / 8;             // 16-byte chunks per row (upper bound)
  constexpr int LD = (kTile * CPR_MAX + 255) / 256;  // chunk loads per thread per operand per tile
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  // (kv head, split, sequence) of this workgroup. Workgroups are handed to the 8 XCDs round-robin in launch order; the
  // kv heads of one token range read neighbouring 256-byte pieces of the same cache rows, and consecutive splits of a
  // sequence neighbouring rows: the launch index is re-ordered so that those run on the same XCD (one L2, and DRAM pages
  // opened once instead of once per XCD).
  int wg = blockIdx.y + gridDim.y * blockIdx.z;
  {
    const int total = gridDim.y * gridDim.z;
    if ((total & 7) == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);
  }
  const int hk = wg % p.Hk;
  const int split = (wg / p.Hk) % p.splits;
  const int b = wg / (p.Hk * p.splits);
  const int D = p.D, G = p.G;
  const int cpr = D >> 3;                      // valid 16-byte chunks per row
  const int nt_valid = (D + 15) >> 4;

  const int q_begin = cu_q[b];
  const int seqlen_q = cu_q[b + 1] - q_begin;
  // Cache layouts (reference flash_attention.cpp:383, :408-412, :649-653; tests/test_flash_attention.py:855-893): sequence
  // b lives in cache row kv_batch_idx[b] (default b) at cache positions [leftpad_k[b], seq_k[b]): key index i of the
  // attention problem is cache position i + leftpad.
  int seqlen_k, k_begin = 0, cache_row = b, leftpad = 0;
  if (p.paged) {
    if (p.kv_batch_idx != nullptr) cache_row = __builtin_amdgcn_readfirstlane(p.kv_batch_idx[b]);  // (uniform, and the
    if (p.leftpad_k != nullptr) leftpad = __builtin_amdgcn_readfirstlane(p.leftpad_k[b]);          // compiler must know)
    seqlen_k = seq_k[b] - leftpad;
    seqlen_k = seqlen_k > 0 ? seqlen_k : 0;
  } else {
    k_begin = seq_k[b];
    seqlen_k = seq_k[b + 1] - k_begin;
  }
  const int rows_total = seqlen_q * G;
  const int row0 = blockIdx.x * kBlockM;
  if (row0 >= rows_total) return;
  const int shift = seqlen_k - seqlen_q;       // bottom-right alignment

  // ---- this wave's 16 packed rows; lane l15 <-> row
  const int my_row = row0 + wave * kRowsPerWave + l15;
  const bool row_ok = my_row < rows_total;
  const int my_qpos = row_ok ? my_row / G : 0;
  const int my_head = hk * G + (row_ok ? my_row % G : 0);
  const int q_abs = my_qpos + shift;
  // bounds over the 16 rows of this wave (wave-uniform), for the interior-tile test of the main loop
  const int wrow_first = row0 + wave * kRowsPerWave, wrow_last = wrow_first + kRowsPerWave - 1;
  const bool wave_rows_ok = wrow_last < rows_total;
  const bool wave_active = wrow_first < rows_total;
  const int wave_qabs_lo = wrow_first / G + shift, wave_qabs_hi = (wrow_last < rows_total ? wrow_last : rows_total - 1) / G + shift;

  // ---- kv range visible to this workgroup (union over its rows), then this split's share of it
  const int last_row = (row0 + kBlockM < rows_total ? row0 + kBlockM : rows_total) - 1;
  const int qpos_lo = row0 / G, qpos_hi = last_row / G;
  int kv_hi = seqlen_k;
  if (p.causal_right >= 0) {
    const int lim = qpos_hi + shift + p.causal_right + 1;
    kv_hi = lim < kv_hi ? lim : kv_hi;
  }
  int kv_lo = 0;
  if (p.window_left >= 0) {
    const int lim = qpos_lo + shift - p.window_left;
    kv_lo = lim > 0 ? lim : 0;
  }
  if (kv_hi < 0) kv_hi = 0;
  int t_lo = kv_lo / kTile, t_hi = (kv_hi + kTile - 1) / kTile;  // tile range [t_lo, t_hi)
  if (t_hi < t_lo) t_hi = t_lo;
  if (p.splits > 1) {
    const int per = (t_hi - t_lo + p.splits - 1) / p.splits;
    const int a = t_lo + split * per;
    const int e = a + per;
    t_lo = a < t_hi ? a : t_hi;
    t_hi = e < t_hi ? e : t_hi;
  }
  const int n_tiles = t_hi - t_lo;

  // k permutation inside a 32-deep MFMA step and token permutation inside a 16-token tile (see mla_decode.hip)
  const int pig = (0x2130 >> (4 * g4)) & 3;
  const int tau = (l15 & 3) | (((l15 >> 2) & 1) << 3) | (((l15 >> 3) & 1) << 2);

  // ---- Q^T fragments
  v8s qf[KS];
  {
    const T* qrow = q + (int64_t)(q_begin + my_qpos) * p.q_s0 + (int64_t)my_head * p.q_s1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 32 * ks + 8 * pig;
      const v8s v = *reinterpret_cast<const v8s*>(qrow + (d0 < D ? d0 : 0));  // (unconditional load, then select)
      const v8s zero = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = (row_ok && d0 < D) ? v : zero;
    }
  }

  // ---- staging: chunk id c = tid + 256 i  ->  (token row = c / cpr, chunk = c % cpr)
  // kSets register sets: the loads of tiles t+2 .. t+kSets are in flight while tile t is multiplied and tile t+1 waits to be written
  // to LDS (with one set a decode-sized tile had ~0.3 us of work to hide ~2 us of latency behind)
  struct Stage {
    v4i k[LD], v[LD];
  };
  constexpr int kSets = DKP > 256 ? 1 : DKP > 128 ? 2 : 3;  // (fewer for the large head dims: registers)
  Stage sreg[kSets];
  // One address formula for the three layouts: page * s0 + position-in-page * s1 + base, with the page term switched off
  // (stride 0, ids read from a dummy word) when there is no table. Positions past the end are clamped to the last key -
  // their scores are masked and a real V row times weight 0 is 0 - so a tile's loads are unconditional, and the page ids
  // of a tile are fetched kSets tiles before its loads: vmcnt retires in order, so waiting for a fetch issued one tile
  // ahead would also wait for every K / V load in front of it and empty the ring; at distance kSets those have landed.
  const bool use_table = p.paged == 1;
  const int32_t* pg_src = use_table ? page_table + (int64_t)cache_row * p.table_stride : cu_q + b;
  const int pos_mask = use_table ? (1 << p.page_shift) - 1 : -1;
  const int pos_shift = use_table ? p.page_shift : 31;  // (position >> 31 = 0)
  const int pos_base = p.paged ? leftpad : k_begin;
  // (strides are below 2^31 elements - checked on the host - so each product is one v_mad_u64_u32)
  const uint32_t kpg = use_table ? (uint32_t)p.k_s0 : 0u, vpg = use_table ? (uint32_t)p.v_s0 : 0u;          // page stride
  const uint32_t kst = (uint32_t)(p.paged ? p.k_s1 : p.k_s0), vst = (uint32_t)(p.paged ? p.v_s1 : p.v_s0);  // token stride
  const int64_t kbase_off = p.paged == 2 ? (int64_t)cache_row * p.k_s0 + (int64_t)hk * p.k_s2
                                         : (int64_t)hk * (p.paged ? p.k_s2 : p.k_s1);
  const int64_t vbase_off = p.paged == 2 ? (int64_t)cache_row * p.v_s0 + (int64_t)hk * p.v_s2
                                         : (int64_t)hk * (p.paged ? p.v_s2 : p.v_s1);
  const int last_key = seqlen_k - 1;
  int srow[LD], sch[LD];  // this thread's (token row, chunk) of a tile; rows >= kTile (odd head dims) load row kTile-1, unused
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    const int c = tid + 256 * i;
    srow[i] = c / cpr;
    sch[i] = c - srow[i] * cpr;
  }
  struct Pages {
    int pg[LD];
  };
  auto fetch_pages = [&](int t) -> Pages {
    Pages r;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      int pos = t * kTile + (srow[i] < kTile ? srow[i] : kTile - 1);
      pos = pos < last_key ? pos : last_key;
      r.pg[i] = pg_src[(pos + pos_base) >> pos_shift];
    }
    return r;
  };
  auto issue_loads = [&](int t, const Pages& pages, Stage& sr) {
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      int pos = t * kTile + (srow[i] < kTile ? srow[i] : kTile - 1);
      pos = pos < last_key ? pos : last_key;
      const uint32_t cp = (uint32_t)((pos + pos_base) & pos_mask), pg = (uint32_t)pages.pg[i];
      const int64_t koff = (int64_t)((uint64_t)pg * kpg + ((uint64_t)cp * kst + (uint64_t)(kbase_off + sch[i] * 8)));
      const int64_t voff = (int64_t)((uint64_t)pg * vpg + ((uint64_t)cp * vst + (uint64_t)(vbase_off + sch[i] * 8)));
      if constexpr (KV8 == 0) {
        sr.k[i] = *reinterpret_cast<const v4i*>(kcache + koff * 2);
        sr.v[i] = *reinterpret_cast<const v4i*>(vcache + voff * 2);
      } else {  // 8 bytes = the 8 elements of this chunk
        const uint2 k8 = *reinterpret_cast<const uint2*>(kcache + koff);
        const uint2 v8 = *reinterpret_cast<const uint2*>(vcache + voff);
        sr.k[i] = (v4i){(int)k8.x, (int)k8.y, 0, 0};
        sr.v[i] = (v4i){(int)v8.x, (int)v8.y, 0, 0};
      }
    }
  };
  // fp8 cache: 8 bytes -> 8 elements of T (every e4m3 / e5m2 value is exact in bf16 and fp16): one
  // v_cvt_scalef32_pk_{bf16,f16}_{fp8,bf8} per pair, unit scale (cvt_pk_f32 + two f32 -> T conversions + packing before)
  auto widen = [&](const v4i& x) -> v4i {
    if constexpr (KV8 == 0) {
      return x;
    } else {
      typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
      typedef _Float16 v2h_ __attribute__((ext_vector_type(2)));
      v4i r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if constexpr (std::is_same<T, bf16>::value && KV8 == 1) {
          r[2 * h] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x[h], 1.0f, false));
          r[2 * h + 1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x[h], 1.0f, true));
        } else if constexpr (std::is_same<T, bf16>::value) {
          r[2 * h] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(x[h], 1.0f, false));
          r[2 * h + 1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(x[h], 1.0f, true));
        } else if constexpr (KV8 == 1) {
          r[2 * h] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x[h], 1.0f, false));
          r[2 * h + 1] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x[h], 1.0f, true));
        } else {
          r[2 * h] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(x[h], 1.0f, false));
          r[2 * h + 1] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(x[h], 1.0f, true));
        }
      }
      return r;
    }
  };
  auto write_lds = [&](int slot, const Stage& sr) {
    char* base = smem + slot * SLOT;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int row = srow[i], ch = sch[i];
      if (row < kTile) {
        const int off = (ch >> 4) * (kTile * 256) + row * 256 + (((ch & 15) ^ sw_main(row)) << 4);
        *reinterpret_cast<v4i*>(base + off) = widen(sr.k[i]);
        *reinterpret_cast<v4i*>(base + TILE_BYTES + off) = widen(sr.v[i]);
      }
    }
  };
  // zero the padded head-dim chunks of every K slot once (they meet zero Q fragments, but must not be NaN)
  if (cpr < CPR_MAX) {
    for (int c = tid; c < kSlots * kTile * (CPR_MAX - cpr); c += 256) {
      const int slot = c / (kTile * (CPR_MAX - cpr));
      const int r = c - slot * (kTile * (CPR_MAX - cpr));
      const int row = r / (CPR_MAX - cpr), ch = cpr + r % (CPR_MAX - cpr);
      const int off = (ch >> 4) * (kTile * 256) + row * 256 + (((ch & 15) ^ sw_main(row)) << 4);
      *reinterpret_cast<v4i*>(smem + slot * SLOT + off) = (v4i){0, 0, 0, 0};
      *reinterpret_cast<v4i*>(smem + slot * SLOT + TILE_BYTES + off) = (v4i){0, 0, 0, 0};
    }
  }

  // ---- per-lane LDS read bases (the second 16-token tile is +4096 bytes)
  const int kbase = 256 * tau + 16 * (pig ^ sw_main(tau));
  int vbase0;
  {
    const int qq = l15 >> 2, pp = l15 & 3;
    const int r = 8 * (g4 & 1) + 4 * (g4 >> 1) + qq;
    vbase0 = 256 * r + 16 * ((pp >> 1) ^ sw_main(r)) + 8 * (pp & 1);
  }

  v4f o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float log2e = 1.4426950408889634f;
  const float kd = (KV8 != 0 && p.k_descale) ? p.k_descale[0] : 1.f, vd = (KV8 != 0 && p.v_descale) ? p.v_descale[0] : 1.f;
  const float scale = p.scale * kd;  // the K descale multiplies every logit
  const float sc2 = scale * log2e;
  __shared__ float xch_all[kWaves * 16];
  float* xch = xch_all + wave * 16;

  Pages pgring[kSets];  // page ids of tiles t + kSets .. t + 2 kSets - 1
  if (n_tiles > 0) {
    Pages first[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) first[j] = fetch_pages(t_lo + j < t_hi ? t_lo + j : t_hi - 1);
#pragma unroll
    for (int j = 0; j < kSets; ++j) pgring[j] = fetch_pages(t_lo + kSets + j < t_hi ? t_lo + kSets + j : t_hi - 1);
#pragma unroll
    for (int j = 0; j < kSets; ++j) issue_loads(t_lo + j < t_hi ? t_lo + j : t_hi - 1, first[j], sreg[j]);
    write_lds(0, sreg[0]);
  }
  __syncthreads();

  // tile i: multiply from LDS slot i % kSlots; registers sreg[i % kSets] (tile i, already in LDS) take the loads of tile
  // i + kSets; the set of tile i + 1 goes to LDS at the end. Unrolled kSets times so that the set index is static.
  auto body = [&](int i, auto parc) {
    constexpr int par = decltype(parc)::value;
    const int t = t_lo + i;
    const int slot = i % kSlots;
    {  // no branch around these (past this workgroup's range they re-read its last tile: cache hits, never used): a
       // conditional load is merged with the old registers through copies that wait for it on the spot
      const int tl = t_hi - 1;
      issue_loads(t + kSets < tl ? t + kSets : tl, pgring[par % kSets], sreg[par % kSets]);
      // (after the last use of the old ids, so that both can live in the same registers: no copy at the loop end)
      pgring[par % kSets] = fetch_pages(t + 2 * kSets < tl ? t + 2 * kSets : tl);
    }
    if (wave_active) {  // (decode: the rows of a kv head fill one wave; the others only stage tiles)
      const char* kb = smem + slot * SLOT;
      const char* vb = kb + TILE_BYTES;
      v4f s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = (ks >> 2) * (kTile * 256) + (kbase ^ ((ks & 3) << 6));
        const v8s a0 = *reinterpret_cast<const v8s*>(kb + off);
        const v8s a1 = *reinterpret_cast<const v8s*>(kb + off + 4096);
        s0 = M::run(a0, qf[ks], s0);
        s1 = M::run(a1, qf[ks], s1);
      }
      // Interior tiles (every key visible to every row of this wave, no softcap) take a short path: the softmax
      // VALU work, not the MFMAs, bounds this kernel at head dim 128 (about 100 VALU ops per 16 x 32 logits with
      // the masks against 16 MFMAs), so the mask arithmetic is skipped wherever a uniform test allows it.
      const int tb = t * kTile + 8 * (g4 & 1) + 4 * (g4 >> 1);
      bool interior = p.softcap <= 0.f && wave_rows_ok && (t * kTile + kTile <= seqlen_k);
      if (p.causal_right >= 0) interior = interior && (t * kTile + kTile - 1 <= wave_qabs_lo + p.causal_right);
      if (p.window_left >= 0) interior = interior && (t * kTile >= wave_qabs_hi - p.window_left);
      float m_new, m_use, alpha, psum = 0.f;
      v8s pf;
      if (interior) {
        float mt = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        m_new = fmaxf(m_run, mt * scale);  // scale > 0: max commutes with it
        m_use = m_new;
        alpha = __builtin_amdgcn_exp2f((m_run - m_use) * log2e);
        const float mneg = -m_use * log2e;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], sc2, mneg));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], sc2, mneg));
          psum += p0 + p1;
          pf[r] = M::cvt(p0);
          pf[4 + r] = M::cvt(p1);
        }
      } else {
        // logits in natural units, then masks
        float z0[4], z1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = s0[r] * scale, c = s1[r] * scale;
          if (p.softcap > 0.f) {
            a = p.softcap * tanhf(a / p.softcap);
            c = p.softcap * tanhf(c / p.softcap);
          }
          const int k0 = tb + r, k1 = tb + 16 + r;
          bool m0 = !row_ok || k0 >= seqlen_k, m1 = !row_ok || k1 >= seqlen_k;
          if (p.causal_right >= 0) { m0 |= k0 > q_abs + p.causal_right; m1 |= k1 > q_abs + p.causal_right; }
          if (p.window_left >= 0) { m0 |= k0 < q_abs - p.window_left; m1 |= k1 < q_abs - p.window_left; }
          z0[r] = m0 ? -INFINITY : a;
          z1[r] = m1 ? -INFINITY : c;
        }
        float mt = fmaxf(fmaxf(fmaxf(z0[0], z0[1]), fmaxf(z0[2], z0[3])), fmaxf(fmaxf(z1[0], z1[1]), fmaxf(z1[2], z1[3])));
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        m_new = fmaxf(m_run, mt);
        // rows that have seen no visible key yet keep m = -inf; use 0 as the reference point to avoid inf - inf
        m_use = m_new == -INFINITY ? 0.f : m_new;
        alpha = __builtin_amdgcn_exp2f((m_run - m_use) * log2e);  // m_run = -inf -> 0
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = __builtin_amdgcn_exp2f((z0[r] - m_use) * log2e);
          const float p1 = __builtin_amdgcn_exp2f((z1[r] - m_use) * log2e);
          psum += p0 + p1;
          pf[r] = M::cvt(p0);
          pf[4 + r] = M::cvt(p1);
        }
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (__any(alpha != 1.0f)) {
        if (lane < 16) xch[lane] = alpha;
        const v4f a4 = *reinterpret_cast<const v4f*>(xch + 4 * g4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          o[nt][0] *= a4[0]; o[nt][1] *= a4[1]; o[nt][2] *= a4[2]; o[nt][3] *= a4[3];
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt < nt_valid) {
          const char* a = vb + (nt >> 3) * (kTile * 256) + (vbase0 ^ ((nt & 7) << 5));
          const v4s v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)SGLK_LDS(a));
          const v4s v1 =
              __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)SGLK_LDS(a + 4096));
          v8s vf;
          vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
          vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
          o[nt] = M::run(pf, vf, o[nt]);
        }
      }
    }

    // (barriers that order LDS only: __syncthreads() would also wait for the K / V loads in flight)
    if (kSlots == 2) lds_barrier();                     // two slots: everyone must be done with the slot being refilled
    if (i + 1 < n_tiles) write_lds((i + 1) % kSlots, sreg[(par + 1) % kSets]);  // three slots: last read two iterations ago
    lds_barrier();
  };
  // whole groups of kSets tiles in a branch-free loop body (a conditional iteration merges the register sets of its two
  // paths through copies, and a copy of a register waits for the load into it - with vmcnt in order that is a wait for
  // every load in flight), then the remaining tiles
  int i_main = 0;
  for (; i_main + kSets <= n_tiles; i_main += kSets) {
    body(i_main, std::integral_constant<int, 0>{});
    if constexpr (kSets >= 2) body(i_main + 1, std::integral_constant<int, 1>{});
    if constexpr (kSets >= 3) body(i_main + 2, std::integral_constant<int, 2>{});
  }
  if constexpr (kSets >= 2) {
    if (i_main < n_tiles) body(i_main, std::integral_constant<int, 0>{});
  }
  if constexpr (kSets >= 3) {
    if (i_main + 1 < n_tiles) body(i_main + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const bool final_pass = p.splits == 1;
  const float m_fin = m_run;
  // natural-log LSE of this pass; -inf when nothing was visible
  float lse_val = (l_tot > 0.f && m_fin != -INFINITY) ? m_fin + logf(l_tot) : -INFINITY;
  if (final_pass && p.sinks != nullptr && row_ok) {
    // the sink logit joins the softmax denominator (reference attention_ref: an extra score column with no value)
    const float sk = p.sinks[my_head];
    const float m2 = fmaxf(m_fin, sk);
    const float l2 = l_tot * __builtin_amdgcn_exp2f((m_fin - m2) * log2e) + __builtin_amdgcn_exp2f((sk - m2) * log2e);
    lse_val = m2 + logf(l2);
    // O is relative to m_fin: its normaliser is l + exp(sink - m_fin) (inf -> output 0)
    l_tot = (m_fin == -INFINITY) ? INFINITY : l_tot + __builtin_amdgcn_exp2f((sk - m_fin) * log2e);
  }
  const float inv_l = (l_tot > 0.f && l_tot < INFINITY) ? vd / l_tot : 0.f;  // (V descale folded in)
  if (lane < 16) xch[lane] = inv_l;
  const v4f i4 = *reinterpret_cast<const v4f*>(xch + 4 * g4);

  // O tile nt: lane holds dim 16 nt + l15 of rows 4 g4 + r of this wave
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + wave * kRowsPerWave + 4 * g4 + r;
    if (row >= rows_total) continue;
    const int qpos = row / G, head = hk * G + row % G;
    const int64_t tok = q_begin + qpos;
    if (final_pass) {
      T* orow = (T*)p.out + tok * p.o_s0 + (int64_t)head * p.o_s1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int d = nt * 16 + l15;
        if (d < D) orow[d] = (T)(o[nt][r] * i4[r]);
      }
    } else {
      float* orow = p.part_o + (((int64_t)split * p.total_q + tok) * p.Hq + head) * D;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int d = nt * 16 + l15;
        if (d < D) orow[d] = o[nt][r] * i4[r];
      }
    }
  }
  if (row_ok && g4 == 0) {
    const float lse = lse_val;
    const int64_t tok = q_begin + my_qpos;
    if (final_pass) p.lse[(int64_t)my_head * p.total_q + tok] = lse;
    else p.part_lse[((int64_t)split * p.Hq + my_head) * p.total_q + tok] = lse;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Prefill kernel (head dims 64 / 128 / 256, 16-bit KV; everything else of the contract above, softcap included): the kernel above gives a
// wave 16 packed rows and 32-token tiles, so every K / V fragment it reads from LDS feeds one MFMA and every tile costs
// a barrier per 16 MFMAs of a wave: 267 TFLOP/s on the causal 16 x 4096 Llama-3-8B prefill (0.11 of the bf16 MFMA
// peak), LDS-read bound. Here a wave takes 32 packed rows and a tile is 64 tokens:
//   * S^T = K . Q^T and O^T = V^T . P^T with v_mfma_f32_32x32x16: lane l owns query row l % 32 in BOTH products
//     (score and output columns), so the running reference / sum / rescale factors are per-lane scalars and P never moves
//     between lanes: a lane's 16 scores of a 32-token block, taken 8 at a time, ARE its P operand for the k-slot
//     order tau(s, u, e) = 32 (s / 2) + 16 (s % 2) + 8 (e / 4) + 4 u + e % 4 (u = lane / 32), and V^T is read in that
//     token order with the hardware transpose read (4 tokens x 16 dims per 16 lanes);
//   * per wave and tile 16 + 16 MFMAs of 32 cycles against 24 KiB of LDS reads (the 16-row kernel: 32 KiB for half the
//     work);
//   * K / V tiles go global -> LDS by LDS-DMA (round 4; see "staging" in the kernel), two LDS buffers, one barrier per 64
//     tokens; LDS images are 256-byte rows with 16-byte chunk c of row r at c ^ (r & 15) for K (ds_read_b128 of 16 different
//     rows is conflict-free) and c ^ ((r & 3) << 2) for V (the 4-row x 64-byte footprint of a transpose read covers all
//     banks once);
//   * causal / window masks only on tiles that are not fully visible to the wave's 32 rows; workgroups of later (longer)
//     row blocks are dispatched first;
//   * NW = 4 waves (128 rows) per workgroup, TWO workgroups per CU (round 4; 8 waves x 32 rows, one workgroup per CU, before:
//     the staging registers the DMA freed - 32 per thread - were what kept a second workgroup out). One workgroup's Q
//     loads, first tiles, barrier waits and output stores now run under the other's tiles: causal 16 x 4096 at d = 128
//     879 - 886 -> 909 - 923 TFLOP/s, at d = 64 684 - 707 -> 827 - 835 (same box, interleaved runs), and with the DMA pieces
//     between the softmax's steps (see "Spread" in the kernel) and the matrix phases at priority 1: 944 - 958 / 838 - 857; a
//     launch of exactly one round of workgroups (q = 128 chunks over 4096 keys: nothing to overlap, and no phase skew between
//     the two waves of a SIMD) went 892 - 923 -> 853 - 860 -> 876 - 890 -> 935 over these steps.
// Round 4, measured and dropped: the weights of a block formed in the gaps of the next block's MFMAs (840
// against 910 TFLOP/s - a wave's vector and matrix instructions do not overlap on this part, interleaved or not, and
// the interleaved form waits for every fragment read); the first fragments of a phase requested ahead of the scalar work
// in front of it (d = 128 unchanged, d = 64 -15 %); MB = 2 (64 rows per wave, one workgroup per CU, the compiler allocating the
// 512 registers): 353 TFLOP/s - ~170 v_accvgpr moves per tile and nobody to overlap with; that form needs asm-owned AGPR
// accumulators and a hand-placed schedule (mla_rows128x_kernel). MB = 2 stays instantiated in the diagnostic build only.
constexpr int kPTile = 64;  // (a workgroup takes 32 NW packed rows)


template <typename T>
struct Mfma32;
template <>
struct Mfma32<bf16> {
  static __device__ __forceinline__ v16f run(const v8s& a, const v8s& b, const v16f& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
  }
};
template <>
struct Mfma32<f16> {
  static __device__ __forceinline__ v16f run(const v8s& a, const v8s& b, const v16f& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
};

// Head dim 64 (round 3; reference instantiations FMHAPrefillXe20.cmake): the same kernel with 128-byte LDS rows - 8 chunks per
// row, one staging load per thread, tile and operand; K chunk c of row r at c ^ ((r >> 1) & 7) (rows two apart share their
// banks), V chunk c at c ^ (((r >> 1) & 1) << 2); 8 + 8 MFMAs per wave and tile against the same softmax work.
// Head dims 96 / 192 (round 5; reference instantiations FMHAPrefillXe20.cmake:30-54): DA = the head dim, D = 128 / 256 the LDS
// image it lives in - 12 (24) of a row's 16 (32) chunk positions are real, the others are filled with a second copy of real chunks
// (the DMA lane fetches chunk c - (D - DA) / 8 instead: no read past a row's end) and never read: Q K^T runs DA / 16 k-steps, P V
// DA / 32 dim blocks - three quarters of the matrix work of the image's size.
template <typename T, int D, int NW, int MB, int KV8 = 0, int DA = D>  // NW waves of MB 32-row blocks; KV8: 0 16-bit cache, 1 e4m3, 2 e5m2
__global__ __launch_bounds__(64 * NW, (NW == 4 && MB == 1 && D <= 128) ? 2 : 1) void attn_prefill_kernel(AttnParams p, const T* __restrict__ q,
                                                           const char* __restrict__ kcache, const char* __restrict__ vcache,
                                                           const int32_t* __restrict__ cu_q, const int32_t* __restrict__ seq_k,
                                                           const int32_t* __restrict__ page_table) {
  using M = Mfma<T>;
  using M32 = Mfma32<T>;
  constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, TILE_BYTES = kPTile * ROWB;
  static_assert(D == 64 || D == 128 || D == 256, "head dims with a power-of-two number of 16-byte chunks per row");
  static_assert(DA == D || (KV8 == 0 && DA % 32 == 0 && DA < D && 2 * DA > D), "a head dim inside the next image size");
  constexpr int KSA = DA / 16, DBA = DA / 32;  // k-steps of Q K^T and dim blocks of P V that are real
  constexpr int kPBlockM = 32 * MB * NW, NTH = 64 * NW;
  constexpr int CPR = D / 8, RPP = NTH / CPR, NCH = kPTile / RPP;  // 16-byte chunks per row, rows per pass, staging loads per thread
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2][K tile, V tile]
  // probe (diagnostic build): 64 no output stores, 128 no Q loads (the probes of the tile loop - no softmax, no MFMAs, no
  // staging, no barrier, no fragment reads - were used once and removed: DESIGN 4.11)
#ifdef SGLK_PROBES
  const int probe = p.probe;
#else
  constexpr int probe = 0;
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, u = lane >> 5;
  // launch index -> (row block, kv head, sequence), re-ordered so that the row blocks of one (sequence, kv head) - which
  // stream the same K / V - run on the same XCD (workgroups go to the 8 XCDs round-robin in launch order)
  int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {
    const int total = gridDim.x * gridDim.y * gridDim.z;
    if ((total & 7) == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);
  }
  const int bx = wg % gridDim.x;
  // (grid.y = kv heads x KV splits - round 5, late: a chunk of a single long sequence is a handful of row blocks, bs 1 x 128 queries
  //  over 32768 keys ran 654 us on 32 workgroups; a split writes its normalised partial result and log-sum-exp for the reduce
  //  kernel exactly as the decode kernel does)
  const int hk = ((wg / gridDim.x) % gridDim.y) % p.Hk, split = ((wg / gridDim.x) % gridDim.y) / p.Hk;
  const int b = wg / (gridDim.x * gridDim.y), G = p.G;

  const int q_begin = cu_q[b];
  const int seqlen_q = cu_q[b + 1] - q_begin;
  int seqlen_k, k_begin = 0, cache_row = b, leftpad = 0;
  if (p.paged) {
    // (loads through pointers inside the parameter struct are vector loads: without the readfirstlane the compiler treats
    // everything derived from them - the tile range, the loop itself - as divergent)
    if (p.kv_batch_idx != nullptr) cache_row = __builtin_amdgcn_readfirstlane(p.kv_batch_idx[b]);
    if (p.leftpad_k != nullptr) leftpad = __builtin_amdgcn_readfirstlane(p.leftpad_k[b]);
    seqlen_k = seq_k[b] - leftpad;
    seqlen_k = seqlen_k > 0 ? seqlen_k : 0;
  } else {
    k_begin = seq_k[b];
    seqlen_k = seq_k[b + 1] - k_begin;
  }
  const int rows_total = seqlen_q * G;
  const int nblk = (rows_total + kPBlockM - 1) / kPBlockM;
  if (bx >= nblk) return;
  const int row0 = (nblk - 1 - bx) * kPBlockM;  // longest rows first
  const int shift = seqlen_k - seqlen_q;

  // (a lane owns row l31 of each of the wave's MB blocks)
  int my_row[MB], my_qpos[MB], my_head[MB], q_abs[MB];
  bool row_ok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    my_row[mb] = row0 + wave * (32 * MB) + 32 * mb + l31;
    row_ok[mb] = my_row[mb] < rows_total;
    my_qpos[mb] = row_ok[mb] ? my_row[mb] / G : 0;
    my_head[mb] = hk * G + (row_ok[mb] ? my_row[mb] % G : 0);
    q_abs[mb] = my_qpos[mb] + shift;
  }
  const int wrow_first = row0 + wave * (32 * MB), wrow_last = wrow_first + 32 * MB - 1;
  const bool wave_rows_ok = wrow_last < rows_total;
  const int wave_qabs_lo = wrow_first / G + shift, wave_qabs_hi = (wrow_last < rows_total ? wrow_last : rows_total - 1) / G + shift;

  const int last_row = (row0 + kPBlockM < rows_total ? row0 + kPBlockM : rows_total) - 1;
  const int qpos_lo = row0 / G, qpos_hi = last_row / G;
  int kv_hi = seqlen_k;
  if (p.causal_right >= 0) {
    const int lim = qpos_hi + shift + p.causal_right + 1;
    kv_hi = lim < kv_hi ? lim : kv_hi;
  }
  int kv_lo = 0;
  if (p.window_left >= 0) {
    const int lim = qpos_lo + shift - p.window_left;
    kv_lo = lim > 0 ? lim : 0;
  }
  if (kv_hi < 0) kv_hi = 0;
  int t_lo = __builtin_amdgcn_readfirstlane(kv_lo / kPTile), t_hi = __builtin_amdgcn_readfirstlane((kv_hi + kPTile - 1) / kPTile);
  if (t_hi < t_lo) t_hi = t_lo;
  if (p.splits > 1) {
    const int per = (t_hi - t_lo + p.splits - 1) / p.splits;
    const int a = t_lo + split * per, e = a + per;
    t_lo = a < t_hi ? a : t_hi;
    t_hi = e < t_hi ? e : t_hi;
  }
  const int n_tiles = t_hi - t_lo;  // (uniform, and the compiler has to know: the tile loop is then a scalar loop)

  // ---- Q^T fragments (B operand of K . Q^T): lane supplies q[row l31][16 ks + 8 u .. + 8)
  v8s qf[MB][KS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const T* qrow = q + (int64_t)(q_begin + my_qpos[mb]) * p.q_s0 + (int64_t)my_head[mb] * p.q_s1;
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks) {
      v8s v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row_ok[mb] && !(probe & 128)) v = *reinterpret_cast<const v8s*>(qrow + 16 * ks + 8 * u);
      qf[mb][ks] = v;
    }
  }

  // ---- staging by LDS-DMA (round 4; before: global -> registers -> ds_write, 32 staging registers per thread, which kept
  // the kernel at one 8-wave workgroup per CU - nothing ran under a workgroup's Q loads, first tiles, barriers and output
  // stores: with every phase of the loop switched off the launch still took 0.43 of its time). A tile is 2 x 64 rows of ROWB
  // bytes = 1-KiB pieces of RP rows; wave w moves pieces w PPW .. w PPW + PPW - 1 of K and of V, one
  // global_load_lds_dwordx4 each: lane -> (row r = lane / CPR of the piece, LDS chunk position c = lane % CPR), and since the
  // DMA writes lanes to consecutive LDS addresses the swizzle sits on the GLOBAL side: the lane fetches chunk c ^ key(row).
  // One address formula for the three layouts (page * s0 + position-in-page * s1 + base; page stride 0 and a dummy id
  // source without a table); positions past the end are clamped to the last key (their scores are masked, and a real V row
  // times weight 0 is 0). A tile that lies whole inside the sequence, with pieces that do not straddle pages, has one page id
  // and one position for all the rows a wave moves: its address arithmetic is scalar and done once per operand, a piece is
  // that SGPR base plus a loop-invariant 32-bit lane offset (the general form costs ~12 VALU instructions per piece; a base
  // per piece cost the wave 121 scalar instructions per tile - its own issue time, 0.1 of the tile). Page ids are fetched a tile ahead by vector loads (a
  // scalar load would share lgkmcnt with the LDS reads and turn their counted waits into full ones).
  // The DMA instructions are asm text (an LDS-DMA the compiler can see makes it wait vmcnt(0) in front of every LDS read):
  // their completion is the hand-written s_waitcnt vmcnt(0) in front of the tile barrier.
  constexpr int RP = 1024 / ROWB, NPIECE = kPTile / RP, PPW = NPIECE / NW;  // rows per piece, pieces per tile, per wave
  static_assert(PPW >= 1 && PPW * NW == NPIECE, "pieces must divide over the waves");
  const int32_t* table_b = page_table + (p.paged == 1 ? (int64_t)cache_row * p.table_stride : 0);
  const bool use_table = p.paged == 1;
  const int pos_mask = use_table ? (1 << p.page_shift) - 1 : -1;
  const int pos_shift = use_table ? p.page_shift : 31;           // (position >> 31 = 0)
  // (strides are below 2^31 elements - checked on the host - so each product is one v_mad_u64_u32)
  const uint32_t kpg = use_table ? (uint32_t)p.k_s0 : 0u, vpg = use_table ? (uint32_t)p.v_s0 : 0u;  // page stride
  const int pos_base = p.paged ? leftpad : k_begin;
  const uint32_t kst = (uint32_t)(p.paged ? p.k_s1 : p.k_s0), vst = (uint32_t)(p.paged ? p.v_s1 : p.v_s0);  // token stride
  const int64_t kbase = p.paged == 2 ? (int64_t)cache_row * p.k_s0 + (int64_t)hk * p.k_s2
                                     : (int64_t)hk * (p.paged ? p.k_s2 : p.k_s1);
  const int64_t vbase = p.paged == 2 ? (int64_t)cache_row * p.v_s0 + (int64_t)hk * p.v_s2
                                     : (int64_t)hk * (p.paged ? p.v_s2 : p.v_s1);
  const int last_key = seqlen_k - 1;
  // (without a page table the fetch reads cu_q[b], a valid word, and the page stride is 0: no branch)
  const int32_t* pg_src = use_table ? table_b : cu_q + b;
  const int prow = lane / CPR, pch = lane % CPR;  // this lane's row of a piece and its LDS chunk position
  // global 16-byte chunk of the lane for piece i of this wave (K: the key depends on the piece's rows; V: it does not)
  auto k_chunk = [&](int i) -> int {
    const int row = RP * (wave * PPW + i) + prow;
    const int c = pch ^ (D >= 128 ? (row & 15) : ((row >> 1) & 7));
    return c < DA / 8 ? c : c - (D - DA) / 8;  // (DA < D: the positions past the row's end take a second copy of real chunks)
  };
  // (V: the key is (row & 3) << 2 - at d = 128 a piece is four rows and the key a lane constant; at d = 256 a piece is two rows
  //  and the key depends on the piece's parity)
  auto v_chunk = [&](int i) -> int {
    const int row = RP * (wave * PPW + i) + prow;
    const int c = pch ^ (D >= 128 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2));
    return c < DA / 8 ? c : c - (D - DA) / 8;
  };
  // (fast tiles: the RP PPW consecutive rows a wave moves share their page, so one SGPR base serves the wave's pieces and the
  // piece's row offset rides in the lane offset)
  constexpr int WR = RP * PPW;  // rows of a tile moved by one wave
  const bool wave_aligned = (!use_table || ((1 << p.page_shift) % WR == 0 && (pos_base & (WR - 1)) == 0)) &&
                            kst < (1u << 22) && vst < (1u << 22);
  auto tile_fast = [&](int t) -> bool { return wave_aligned && t * kPTile + kPTile - 1 <= last_key; };
  uint32_t voff_k[PPW], voff_v[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    voff_k[i] = ((uint32_t)(RP * i + prow) * kst) * 2u + (uint32_t)(k_chunk(i) << 4);
    voff_v[i] = ((uint32_t)(RP * i + prow) * vst) * 2u + (uint32_t)(v_chunk(i) << 4);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)SGLK_LDS(smem);
  auto dma16 = [&](const char* src, uint32_t lds_dst) {  // per-lane 64-bit address
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_dst) : "memory", "m0");
  };
  auto dma16s = [&](uint32_t voff, const char* sbase, uint32_t lds_dst) {  // SGPR base + 32-bit lane offset
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory", "m0");
  };
  struct Pages { int pg[PPW]; };  // fast tiles: pg[0] = the id of the wave's rows; else: this lane's id of piece i
  auto fetch_pages = [&](int t) -> Pages {
    Pages r = {};
    if (tile_fast(t)) {
      r.pg[0] = pg_src[((t * kPTile + WR * wave + pos_base) >> pos_shift) + (lane >> 6)];  // (lane >> 6 = 0: a vector load)
      return r;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int pos = t * kPTile + RP * (wave * PPW + i) + prow;
      pos = pos < last_key ? pos : last_key;
      r.pg[i] = pg_src[(pos + pos_base) >> pos_shift];
    }
    return r;
  };
  // K and V pieces of tile t into buffer buf
  auto stage_tile = [&](int t, int buf, const Pages& pages) {
    const uint32_t kdst = lds0 + (uint32_t)(buf * TILE_BYTES + wave * PPW * 1024);
    const uint32_t vdst = kdst + 2 * TILE_BYTES;
    if (tile_fast(t)) {
      const uint32_t pg = (uint32_t)__builtin_amdgcn_readfirstlane(pages.pg[0]);
      const uint32_t cp = (uint32_t)((t * kPTile + WR * wave + pos_base) & pos_mask);
      const char* kb_ = kcache + (int64_t)((uint64_t)pg * kpg + ((uint64_t)cp * kst + (uint64_t)kbase)) * 2;
      const char* vb_ = vcache + (int64_t)((uint64_t)pg * vpg + ((uint64_t)cp * vst + (uint64_t)vbase)) * 2;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        dma16s(voff_k[i], kb_, kdst + i * 1024);
        dma16s(voff_v[i], vb_, vdst + i * 1024);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int pos = t * kPTile + RP * (wave * PPW + i) + prow;
      pos = pos < last_key ? pos : last_key;
      const uint32_t cp = (uint32_t)((pos + pos_base) & pos_mask);
      const int64_t ko = (int64_t)((uint64_t)(uint32_t)pages.pg[i] * kpg + ((uint64_t)cp * kst + (uint64_t)kbase));
      const int64_t vo = (int64_t)((uint64_t)(uint32_t)pages.pg[i] * vpg + ((uint64_t)cp * vst + (uint64_t)vbase));
      dma16(kcache + ko * 2 + (k_chunk(i) << 4), kdst + i * 1024);
      dma16(vcache + vo * 2 + (v_chunk(i) << 4), vdst + i * 1024);
    }
  };

  // Fast tiles, spread form: the bases now, piece j (K pieces first) wherever the caller puts it. Stamped (diagnostic build): the
  // 8 pieces of a d = 128 tile issued back to back held the wave for ~1400 cycles per tile - an LDS-DMA instruction issues in
  // ~170 cycles behind another one - a third of the stamped loop; between the softmax's vector instructions the pieces are
  // ~145 cycles apart and the wave does not wait for them.
  struct Spread { const char* kb; const char* vb; uint32_t kdst; };
  auto spread_begin = [&](int t, int buf, const Pages& pages) -> Spread {
    Spread r;
    const uint32_t pg = (uint32_t)__builtin_amdgcn_readfirstlane(pages.pg[0]);
    const uint32_t cp = (uint32_t)((t * kPTile + WR * wave + pos_base) & pos_mask);
    r.kb = kcache + (int64_t)((uint64_t)pg * kpg + ((uint64_t)cp * kst + (uint64_t)kbase)) * 2;
    r.vb = vcache + (int64_t)((uint64_t)pg * vpg + ((uint64_t)cp * vst + (uint64_t)vbase)) * 2;
    r.kdst = lds0 + (uint32_t)(buf * TILE_BYTES + wave * PPW * 1024);
    return r;
  };
  auto spread_piece = [&](const Spread& sp, int j) {  // j < 2 PPW
    if (j < PPW) dma16s(voff_k[j], sp.kb, sp.kdst + j * 1024);
    else dma16s(voff_v[j - PPW], sp.vb, sp.kdst + 2 * TILE_BYTES + (j - PPW) * 1024);
  };

  // ---- fp8 cache (round 5; reference tests/test_flash_attention.py:1691-1704 - before, an fp8 prefill ran on the general 16-row
  // kernel): the LDS images stay 16-bit, so everything below the staging is unchanged. LDS-DMA cannot widen: a tile goes global ->
  // registers (16 bytes = 16 elements per lane and load, a tile ahead) -> v_cvt_scalef32_pk_* -> two ds_write_b128 into the
  // swizzled image behind the P . V phase. A wave stages rows 16 wave .. + 15 of K and of V. The K descale rides in the softmax
  // scale, the V descale in the final normalisation (as in the decode kernel).
  constexpr int C16 = D / 16;             // 16-byte pieces per fp8 row
  constexpr int RPL = 64 / C16;           // rows per load instruction
  constexpr int NL8 = KV8 ? 16 / RPL : 1; // loads per operand, wave and tile
  struct Regs8 {
    v4i k[NL8], v[NL8];
  };
  auto widen8 = [&](int lo, int hi) -> v4i {  // 8 bytes -> 8 elements of T (exact), unit scale
    typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
    typedef _Float16 v2h_ __attribute__((ext_vector_type(2)));
    v4i r = {0, 0, 0, 0};
    const int x[2] = {lo, hi};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (std::is_same<T, bf16>::value && KV8 == 1) {
        r[2 * h] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x[h], 1.0f, true));
      } else if constexpr (std::is_same<T, bf16>::value && KV8 == 2) {
        r[2 * h] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(x[h], 1.0f, true));
      } else if constexpr (KV8 == 1) {
        r[2 * h] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x[h], 1.0f, true));
      } else if constexpr (KV8 == 2) {
        r[2 * h] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(x[h], 1.0f, true));
      }
    }
    return r;
  };
  auto issue8 = [&](int t, Regs8& r) {
#pragma unroll
    for (int j = 0; j < NL8; ++j) {
      int pos = t * kPTile + 16 * wave + RPL * j + lane / C16;
      pos = pos < last_key ? pos : last_key;
      const uint32_t cp = (uint32_t)((pos + pos_base) & pos_mask);
      const uint32_t pg = (uint32_t)pg_src[(pos + pos_base) >> pos_shift];
      const int64_t ko = (int64_t)((uint64_t)pg * kpg + ((uint64_t)cp * kst + (uint64_t)kbase));  // (fp8: elements = bytes)
      const int64_t vo = (int64_t)((uint64_t)pg * vpg + ((uint64_t)cp * vst + (uint64_t)vbase));
      r.k[j] = *reinterpret_cast<const v4i*>(kcache + ko + 16 * (lane % C16));
      r.v[j] = *reinterpret_cast<const v4i*>(vcache + vo + 16 * (lane % C16));
    }
  };
  auto write8 = [&](int buf, const Regs8& r) {
    char* kbw = smem + buf * TILE_BYTES;
    char* vbw = smem + (2 + buf) * TILE_BYTES;
    // (the eight lanes of a ds_write_b128 group hold eight pieces of one row - or four of each of two rows whose keys have the
    //  same parity: the upper four write their odd chunk first, so that every instruction covers all eight chunk residues)
    const int first = (lane >> 2) & 1;
#pragma unroll
    for (int j = 0; j < NL8; ++j) {
      const int row = 16 * wave + RPL * j + lane / C16, ch = 2 * (lane % C16);
      const int kk = D >= 128 ? (row & 15) : ((row >> 1) & 7), vk = D >= 128 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2);
      const v4i k0 = widen8(r.k[j][0], r.k[j][1]), k1 = widen8(r.k[j][2], r.k[j][3]);
      const v4i v0 = widen8(r.v[j][0], r.v[j][1]), v1 = widen8(r.v[j][2], r.v[j][3]);
      char* kr_ = kbw + row * ROWB;
      char* vr_ = vbw + row * ROWB;
      *reinterpret_cast<v4i*>(kr_ + (((ch + first) ^ kk) << 4)) = first ? k1 : k0;
      *reinterpret_cast<v4i*>(kr_ + (((ch + 1 - first) ^ kk) << 4)) = first ? k0 : k1;
      *reinterpret_cast<v4i*>(vr_ + (((ch + first) ^ vk) << 4)) = first ? v1 : v0;
      *reinterpret_cast<v4i*>(vr_ + (((ch + 1 - first) ^ vk) << 4)) = first ? v0 : v1;
    }
  };

  // ---- per-lane LDS read offsets
  // K (A operand of K . Q^T): token row 32 beta + l31, chunk 2 ks + u at position chunk ^ (row & 15)
  const int krow_off = l31 * ROWB, kkey = D >= 128 ? (l31 & 15) : ((l31 >> 1) & 7);  // (32 beta does not change the key)
  // V^T (A operand of V^T . P^T) by transpose reads: 16 lanes fetch 4 tokens x 16 dims; lane -> (token qq, 4-dim quad pp)
  const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3, hh = (lane >> 4) & 1;
  const int vlane_off = (4 * u + qq) * ROWB + 8 * (pp & 1);  // + token group offsets below; row & 3 == qq
  const int vchunk_lo = 2 * hh + (pp >> 1);                  // chunk = 4 db + vchunk_lo, swizzled with qq << 2

  v16f o[MB][DB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int v = 0; v < 16; ++v) o[mb][db][v] = 0.f;
  // Softmax reference m_ref (log2 units, per row): moved only when a tile's maximum passes it by more than kSlack
  // binades, so that the 64-register rescale of O^T is rare; weights are then at most 2^kSlack (exact arithmetic gives
  // the same result for any reference, the final normalisation divides it out).
  constexpr float kSlack = 8.0f;  // (256.0f below is 2^kSlack)
  float m_ref[MB], l_run[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { m_ref[mb] = -INFINITY; l_run[mb] = 0.f; }
  const float log2e = 1.4426950408889634f;
  const float kd8 = (KV8 != 0 && p.k_descale) ? p.k_descale[0] : 1.f, vd8 = (KV8 != 0 && p.v_descale) ? p.v_descale[0] : 1.f;
  const float scale = p.scale * kd8, sc2 = scale * log2e;  // (fp8 cache: the K descale multiplies every logit)

  // Schedule. Per wave and tile j: QK(j) (MFMA), softmax(j) (VALU), PV(j) (MFMA), one workgroup barrier per tile. Up to
  // round 3 a workgroup was 8 waves - two per SIMD, which took the barrier at different places (between QK and softmax /
  // between softmax and PV, told apart by HW_ID.wave_id) so that one started a matrix phase while the other started its
  // softmax: 817 / 942 TFLOP/s causal / full against 783 / 900 without the skew. With four waves a SIMD's two waves belong
  // to two workgroups, which drift apart by themselves.
  // Tile j lives in buffer j & 1. Iteration j: its pieces have landed (vmcnt(0)) and, behind the barrier, everybody's;
  // everybody is also done with tile j - 1, whose buffer takes the pieces of tile j + 1 now - in flight for the whole of
  // iteration j - with the page ids fetched in iteration j - 1; the ids of tile j + 2 are fetched next.
  Pages pg_next = {};
  Regs8 r8 = {};
  if (n_tiles > 0) {
    if constexpr (KV8 != 0) {
      issue8(t_lo, r8);
      write8(0, r8);
    } else {
      stage_tile(t_lo, 0, fetch_pages(t_lo));
      if (n_tiles > 1) pg_next = fetch_pages(t_lo + 1);
    }
  }

#ifdef SGLK_PROBES
  // in-kernel stamps (guide 7, 'In-kernel stamps'): cycle sums of the tile loop's segments in scalar registers, stored once by
  // lane 0 of every wave. Read the SHARES: the fences forbid overlaps the real loop has.
  unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_prev = 0;
  const bool st_on = p.stamps != nullptr;
#define PF_STAMP(k)                                                                            \
  if (st_on) {                                                                                 \
    unsigned long long now_;                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if ((k) >= 0) st_sum[(k) < 0 ? 0 : (k)] += now_ - st_prev;                                 \
    st_prev = now_;                                                                            \
  }
#else
#define PF_STAMP(k)
#endif
  for (int i = 0; i < n_tiles; ++i) {
    const int t = t_lo + i, buf = i & 1;
    PF_STAMP(-1)
    if constexpr (KV8 != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the image was written by ds_write)
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    PF_STAMP(0)
    // the next tile's staging: a fast tile's pieces ride in the softmax of the wave's first row block (below); any other
    // tile's go out together behind QK
    const bool next_tile = i + 1 < n_tiles, spread = KV8 == 0 && next_tile && tile_fast(t + 1);
    const char* kb = smem + buf * TILE_BYTES;
    const char* vb = smem + (2 + buf) * TILE_BYTES;

    // ---- S^T[token, row] = K . Q^T: two 32-token blocks; K fragments two k-steps ahead of their MFMAs
    v16f s0[MB], s1[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int v = 0; v < 16; ++v) { s0[mb][v] = 0.f; s1[mb][v] = 0.f; }
    {
      // K fragments kKA k-steps ahead of their MFMAs (round 4: three k-steps and six V^T fragments ahead measured the same as
      // two and four - 890 - 900 against 897 - 902 TFLOP/s at d = 128: the fragment reads are covered)
      constexpr int kKA = 2;
      v8s ka[kKA][2];
      auto read_k = [&](int ks, v8s (&dst)[2]) {
        const int off = krow_off + (((2 * ks + u) ^ kkey) << 4);
        dst[0] = *reinterpret_cast<const v8s*>(kb + off);
        dst[1] = *reinterpret_cast<const v8s*>(kb + off + 32 * ROWB);
      };
#pragma unroll
      for (int ks = 0; ks < kKA; ++ks) read_k(ks, ka[ks]);
      // (the matrix phases run at priority 1: the partner wave's softmax then takes the issue slots the MFMAs leave - 918 against
      // 908 TFLOP/s at d = 128, interleaved runs of the diagnostic build)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          s0[mb] = M32::run(ka[ks % kKA][0], qf[mb][ks], s0[mb]);
          s1[mb] = M32::run(ka[ks % kKA][1], qf[mb][ks], s1[mb]);
        }
        if (ks + kKA < KSA) read_k(ks + kKA, ka[ks % kKA]);
      }
      // keep that order: the scheduler otherwise sinks every read below the MFMAs in front of it
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * kKA, 0);
#pragma unroll
      for (int ks = 0; ks < KSA - kKA; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MB, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * kKA * MB, 0);
      __builtin_amdgcn_s_setprio(0);
    }

    PF_STAMP(1)
    Spread sp = {};
    if constexpr (KV8 != 0) {
      if (next_tile) issue8(t + 1, r8);  // (widened and written behind the P . V phase)
    } else {
      if (spread) sp = spread_begin(t + 1, buf ^ 1, pg_next);
      else if (next_tile) stage_tile(t + 1, buf ^ 1, pg_next);
      if (next_tile && i + 2 < n_tiles) pg_next = fetch_pages(t + 2);
    }
    PF_STAMP(2)

    // ---- softcap (Gemma-2; round 5 - before, a capped prefill fell through to the general kernel): the raw score s becomes
    // s'' = (cap / scale) tanh(s scale / cap), so that the weights below (y = s'' scale log2e - reference) need nothing else;
    // tanh(x) = 1 - 2 / (exp(2 x) + 1) as one exp2, one rcp, two fmas per score (saturates correctly at +-inf). In front of the
    // masks: a masked score must stay -inf.
    if (p.softcap > 0.f) {  // (uniform)
      const float c1 = 2.0f * sc2 / p.softcap, c2 = p.softcap / scale;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const float e0 = __builtin_amdgcn_exp2f(s0[mb][v] * c1), e1 = __builtin_amdgcn_exp2f(s1[mb][v] * c1);
          s0[mb][v] = c2 - 2.0f * c2 * __builtin_amdgcn_rcpf(e0 + 1.0f);
          s1[mb][v] = c2 - 2.0f * c2 * __builtin_amdgcn_rcpf(e1 + 1.0f);
        }
    }
    // ---- online softmax for row l31 of each block (this lane: tokens 8 (v / 4) + 4 u + v % 4 of each 32-token block)
    // (rows past the block's last one take part with q = 0 - finite scores, finite weights, sums that are never stored - so a
    //  ragged row count does not send every tile of the wave down the masked path: 124 packed rows ran 1.6 x slower than 128)
    bool interior = t * kPTile + kPTile <= seqlen_k;
    if (p.causal_right >= 0) interior = interior && (t * kPTile + kPTile - 1 <= wave_qabs_lo + p.causal_right);
    if (p.window_left >= 0) interior = interior && (t * kPTile >= wave_qabs_hi - p.window_left);
    if (__builtin_amdgcn_readfirstlane(interior ? 0 : 1)) {  // (uniform per wave)
      const int tb = t * kPTile + 4 * u;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int k0 = tb + 8 * (v >> 2) + (v & 3), k1 = k0 + 32;
          bool m0 = !row_ok[mb] || k0 >= seqlen_k, m1 = !row_ok[mb] || k1 >= seqlen_k;
          if (p.causal_right >= 0) { m0 |= k0 > q_abs[mb] + p.causal_right; m1 |= k1 > q_abs[mb] + p.causal_right; }
          if (p.window_left >= 0) { m0 |= k0 < q_abs[mb] - p.window_left; m1 |= k1 < q_abs[mb] - p.window_left; }
          s0[mb][v] = m0 ? -INFINITY : s0[mb][v];
          s1[mb][v] = m1 ? -INFINITY : s1[mb][v];
        }
    }
    // (Round 4: the weights are formed OPTIMISTICALLY against the running reference - y = s * scale - m_ref is one v_fma per
    // score, no separate scaling product - and no maximum is taken in the common case (below); in the slow path the two
    // lanes of a row exchange their maxima with v_permlane32_swap instead of a ds_bpermute (six VALU instructions of index
    // arithmetic and an LDS round trip). 241 -> 153 vector instructions per tile with the scalar staging - and by itself that
    // bought nothing at d = 128 (875 -> 881 TFLOP/s): the tile is not bound by its vector instruction count. Scalar f32
    // instructions on purpose - the file is built with -fno-slp-vectorize: next to MFMAs a v_pk_*_f32 costs more issue time
    // than the two scalar instructions it replaces. No inline-asm VALU on MFMA results: the hazard recogniser cannot see an
    // asm read of a register an MFMA is still writing (the swap below reads VALU results).)
    typedef short v2s_ __attribute__((ext_vector_type(2)));
    int pw[MB][16];  // rounded weights as dwords: block b, tokens 2 j, 2 j + 1 -> pw[8 b + j]; B fragment h = pw[4 h .. 4 h + 3]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float mneg = m_ref[mb] == -INFINITY ? 0.f : -m_ref[mb];
      float psum_a = 0.f, psum_b = 0.f;
      auto weights = [&](bool with_dma) {
        psum_a = 0.f;
        psum_b = 0.f;
        // in steps of two tokens per block (14 vector instructions); with_dma: one DMA piece of the next tile behind every
        // 8 / (2 PPW)-th step, the steps pinned where they are written (opaque copies of the reference in front, of the results
        // behind: the optimiser would otherwise hoist all the fmas above the first piece and sink sums and roundings below the last)
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          float mg = mneg;
          if (with_dma) asm volatile("" : "+v"(mg));
          const int v = 2 * st;
          const float p0a = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[mb][v], sc2, mg));  // -inf stays -inf (scale > 0)
          const float p0b = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[mb][v + 1], sc2, mg));
          const float p1a = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[mb][v], sc2, mg));
          const float p1b = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[mb][v + 1], sc2, mg));
          psum_a += p0a;
          psum_b += p1a;
          psum_a += p0b;
          psum_b += p1b;
          const v2s_ r0 = {M::cvt(p0a), M::cvt(p0b)}, r1 = {M::cvt(p1a), M::cvt(p1b)};
          int k0 = __builtin_bit_cast(int, r0), k1 = __builtin_bit_cast(int, r1);
          if (with_dma) asm volatile("" : "+v"(k0), "+v"(k1), "+v"(psum_a), "+v"(psum_b));
          pw[mb][st] = k0;
          pw[mb][8 + st] = k1;
          constexpr int kNP = 2 * PPW;  // pieces of a tile per wave: 4 / 8 / 16 at d = 64 / 128 / 256, over the eight steps
          if (mb == 0 && with_dma) {
            if constexpr (kNP >= 8) {
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int q_ = 0; q_ < kNP / 8; ++q_) spread_piece(sp, st * (kNP / 8) + q_);
              __builtin_amdgcn_sched_barrier(0);
            } else if (st % (8 / kNP) == 8 / kNP - 1) {
              __builtin_amdgcn_sched_barrier(0);
              spread_piece(sp, st / (8 / kNP));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      };
      if (spread) weights(true);
      else weights(false);
      // No maximum in the common case: a lane whose 32 weights sum to at most 2^kSlack holds none above 2^kSlack, so its row
      // does not ask for a new reference. Otherwise (or while a row has no reference yet: the sum then says nothing, the raw
      // scores may all underflow) the wave takes the slow path - maxima, exchange, move, weights again.
      if (__any(m_ref[mb] == -INFINITY || !(psum_a + psum_b <= 256.0f))) {
        float x0[16], x1[16];  // (formed again, behind an opaque copy: kept from above they would cost the common path 32 registers)
        float mneg_again = mneg;
        asm volatile("" : "+v"(mneg_again));
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          x0[v] = __builtin_fmaf(s0[mb][v], sc2, mneg_again);
          x1[v] = __builtin_fmaf(s1[mb][v], sc2, mneg_again);
        }
        float mt = fmaxf(fmaxf(x0[0], x0[1]), x1[0]);
        mt = fmaxf(mt, x1[1]);
#pragma unroll
        for (int v = 2; v < 16; v += 2) {
          mt = fmaxf(fmaxf(mt, x0[v]), x0[v + 1]);
          mt = fmaxf(fmaxf(mt, x1[v]), x1[v + 1]);
        }
        {
          float ma = mt, mb_ = mt;
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ma), "+v"(mb_));
          mt = fmaxf(ma, mb_);  // (both halves of the wave now hold the row's maximum)
        }
        // mt is relative to the reference (absolute while there is none): move it when a tile passes it by kSlack binades
        const bool moves = m_ref[mb] == -INFINITY ? mt > -INFINITY : mt > kSlack;
        if (__any(moves)) {
          const float m_new = moves ? (m_ref[mb] == -INFINITY ? mt : m_ref[mb] + mt) : m_ref[mb];
          const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m_ref[mb] - m_new);
          m_ref[mb] = m_new;
          l_run[mb] *= alpha;
#pragma unroll
          for (int db = 0; db < DBA; ++db)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[mb][db][v] *= alpha;
          mneg = m_ref[mb] == -INFINITY ? 0.f : -m_ref[mb];
          weights(false);
        }
      }
      l_run[mb] += psum_a + psum_b;
    }

    PF_STAMP(3)

    // ---- O^T[dim, row] += V^T . P^T: k-slot order tau (see above): MFMA s4 takes tokens 32 (s4 / 2) + 16 (s4 % 2) + ...
    // step m = 4 s4 + db (the four accumulators in turn), MB MFMAs per V^T fragment; fragments four steps ahead
    {
      constexpr int kVA = 4;  // V^T fragments in flight (steps ahead)
      v8s vf[kVA];
      auto read_v = [&](int m, v8s& dst) {
        const int s4 = m / DBA, db = m % DBA;
        const int chunk = ((4 * db + vchunk_lo) ^ ((D >= 128 ? qq : (qq >> 1)) << 2)) << 4;
        const char* a = vb + (32 * (s4 >> 1) + 16 * (s4 & 1)) * ROWB + vlane_off + chunk;
        const v4s v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)SGLK_LDS(a));
        const v4s v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)SGLK_LDS(a + 8 * ROWB));
        dst[0] = v0[0]; dst[1] = v0[1]; dst[2] = v0[2]; dst[3] = v0[3];
        dst[4] = v1[0]; dst[5] = v1[1]; dst[6] = v1[2]; dst[7] = v1[3];
      };
      constexpr int NPV = 4 * DBA;
#pragma unroll
      for (int m = 0; m < kVA; ++m) read_v(m, vf[m]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < NPV; ++m) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const v4i w4 = {pw[mb][4 * (m / DBA)], pw[mb][4 * (m / DBA) + 1], pw[mb][4 * (m / DBA) + 2], pw[mb][4 * (m / DBA) + 3]};
          o[mb][m % DBA] = M32::run(vf[m % kVA], __builtin_bit_cast(v8s, w4), o[mb][m % DBA]);
        }
        if (m + kVA < NPV) read_v(m + kVA, vf[m % kVA]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * kVA, 0);
#pragma unroll
      for (int m = 0; m < NPV - kVA; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, MB, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kVA * MB, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (KV8 != 0) {
      if (next_tile) write8(buf ^ 1, r8);  // (buffer buf ^ 1 was last read in the iteration before: free since this one's barrier)
    }
    PF_STAMP(4)
  }
#ifdef SGLK_PROBES
  if (st_on && lane == 0) {
    unsigned long long* dst = p.stamps + ((int64_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NW + wave) * 8;
    for (int k = 0; k < 5; ++k) dst[k] = st_sum[k];
    dst[5] = (unsigned long long)n_tiles;
    dst[6] = (unsigned long long)bx;
  }
#endif
#undef PF_STAMP

  // ---- epilogue (as the kernel above; the two lanes of a row hold partial sums)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const float m_fin = m_ref[mb] == -INFINITY ? -INFINITY : m_ref[mb] * 0.6931471805599453f;  // the reference in natural-log units
    float l_tot = l_run[mb] + __shfl_xor(l_run[mb], 32, 64);
    float lse_val = (l_tot > 0.f && m_fin != -INFINITY) ? m_fin + logf(l_tot) : -INFINITY;
    const bool final_pass = p.splits == 1;  // (a split leaves the sink term to the reduce kernel)
    if (final_pass && p.sinks != nullptr && row_ok[mb]) {
      const float sk = p.sinks[my_head[mb]];
      const float m2 = fmaxf(m_fin, sk);
      const float l2 = l_tot * __builtin_amdgcn_exp2f((m_fin - m2) * log2e) + __builtin_amdgcn_exp2f((sk - m2) * log2e);
      lse_val = m2 + logf(l2);
      l_tot = (m_fin == -INFINITY) ? INFINITY : l_tot + __builtin_amdgcn_exp2f((sk - m_fin) * log2e);
    }
    const float inv_l = ((l_tot > 0.f && l_tot < INFINITY) ? 1.0f / l_tot : 0.f) * vd8;  // (fp8 cache: x the V descale)
    if (row_ok[mb] && !(probe & 64)) {
      const int64_t tok = q_begin + my_qpos[mb];
      if (final_pass) {
        T* orow = (T*)p.out + tok * p.o_s0 + (int64_t)my_head[mb] * p.o_s1;
#pragma unroll
        for (int db = 0; db < DBA; ++db) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {  // dims 32 db + 8 g + 4 u .. + 3
            Vec<T, 4> ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (T)(o[mb][db][4 * g + r] * inv_l);
            store_vec<T, 4>(orow + 32 * db + 8 * g + 4 * u, ov);
          }
        }
        if (u == 0) p.lse[(int64_t)my_head[mb] * p.total_q + tok] = lse_val;
      } else {
        float* orow = p.part_o + (((int64_t)split * p.total_q + tok) * p.Hq + my_head[mb]) * DA;
#pragma unroll
        for (int db = 0; db < DBA; ++db) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            Vec<float, 4> ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = o[mb][db][4 * g + r] * inv_l;
            store_vec<float, 4>(orow + 32 * db + 8 * g + 4 * u, ov);
          }
        }
        if (u == 0) p.part_lse[((int64_t)split * p.Hq + my_head[mb]) * p.total_q + tok] = lse_val;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Decode kernel (head dim 128, 16-bit KV, at most 16 packed rows per (sequence, kv head): seqlen_q * G <= 16; paged with
// pages of >= 32 tokens, or one of the unpaged layouts; everything else of the contract above). In the general kernel
// the rows of a decode step fill one of the four waves: that wave's read - MFMA - exchange - exp - MFMA chain takes ~1.7 us
// per 32-token tile while the other three wait at the tile's barrier (measured: 72 us for the 268 MB of
// BASELINE configs[2] against 52 us with the arithmetic switched off). Here the four waves of a workgroup run
// INDEPENDENTLY over interleaved tiles (wave w: tiles w, w + 4, ..), each with its own running max / sum / output, merged
// once at the end:
//   * K fragments go global -> registers directly in the A-operand layout of S^T = K . Q^T (lane = (token, 8-dim
//     chunk): 16 bytes per lane and load, no LDS); V goes global -> registers -> a wave-private 8 KiB LDS image
//     (the hardware transpose read needs it) with no workgroup barrier anywhere in the loop;
//   * two tiles in flight per wave (32 KiB; 256 KiB per CU at two workgroups per CU);
//   * the page id of a tile is fetched two tiles ahead, in front of the loads of the tile one ahead (vmcnt retires in
//     order: see the general kernel).
// Head dims 64 / 128 / 256 and an fp8 (e4m3 / e5m2) cache (round 3; reference instantiations FMHADecodeXe20.cmake:13-16,
// :62-111): D / 32 k-steps, D / 16 output tiles; the V image is made of 128-dim column blocks of [32 tokens][256 B] with the
// swizzle of the d = 128 image (d = 64 fills half a block); d = 256 takes the whole register file (one workgroup per CU:
// 64 KiB of K / V per wave in flight is still far more than the latency needs). fp8: K bytes go to registers as they are
// and are widened right before their MFMAs (v_cvt_scalef32_pk_*), V is widened on its way into the LDS image; the K
// descale is folded into the softmax scale, the V descale into the final normalisation (as the general kernel does).
// NW waves per workgroup, wave w on the tiles w, w + NW, ..: four. Eight (two chains per SIMD without more KV splits) were
// tried for the half-size tiles of d = 64 and of an fp8 cache, which sit at 0.46 of HBM: 36.1 against 34.5 us at d = 64, 37.6
// against 35.9 with an fp8 cache, 55.2 against 53.0 at d = 128 (round 4, interleaved runs of the diagnostic build) - the
// stream itself runs at 5.8 - 6.5 TB/s, what the short legs feel are ~12 us of fixed cost (first-tile latency chain, the
// wave merge, the partial results and the reduce launch).
// Head dims 96 / 192 (round 5; the reference's paged decode builds them, FMHADecodeXe20.cmake:13-16): DA = the head dim inside the
// d = 128 / 256 form - DA / 32 k-steps of K (loaded straight into registers: nothing extra is fetched), DA / 16 output tiles, and a V
// image whose rows keep 16 (32) chunk positions of which 12 (24) are real: the lanes of the other positions fetch a second copy of
// real chunks (no read past a row's end), an eighth more HBM bytes than the rows hold.
template <typename T, int D, int KV8, int NW, int DA = D>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : (D <= 128 ? 2 : 1))) void attn_decode_kernel(AttnParams p, const T* __restrict__ q,
                                                             const char* __restrict__ kcache, const char* __restrict__ vcache,
                                                             const int32_t* __restrict__ cu_q, const int32_t* __restrict__ seq_k,
                                                             const int32_t* __restrict__ page_table) {
  using M = Mfma<T>;
  static_assert(DA == D || (DA % 32 == 0 && DA < D && 2 * DA > D), "a head dim inside the next form");
  constexpr int KS = DA / 32, NT = DA / 16, NB = (D + 127) / 128, VIMG = NB * kTile * 256;  // V image of a tile: NB x 8 KiB
  constexpr int ES = KV8 ? 1 : 2;  // bytes per cache element
  static_assert(D == 64 || D == 128 || D == 256, "decode kernel: head dims 64, 128, 256");
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [4 waves][2] V images; the merge reuses them

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  int wg = blockIdx.x + gridDim.x * blockIdx.y;  // x = kv head + Hk * split, y = sequence; re-ordered as above
  {
    const int total = gridDim.x * gridDim.y;
    if ((total & 7) == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);
  }
  const int hk = wg % p.Hk;
  const int split = (wg / p.Hk) % p.splits;
  // 17 - 64 packed rows (speculative decoding, short chunks; round 5 - they ran on the general kernel at a quarter of this kernel's
  // rate): row_groups workgroups per (sequence, kv head, split), 16 rows each, every one streaming the split's keys (the repeats
  // come from L2: the groups of a (sequence, head) are neighbours in the launch)
  const int rg = (wg / (p.Hk * p.splits)) % p.row_groups;
  const int b = wg / (p.Hk * p.splits * p.row_groups);
  const int G = p.G;

  const int q_begin = cu_q[b];
  const int seqlen_q = cu_q[b + 1] - q_begin;
  int seqlen_k, k_begin = 0, cache_row = b, leftpad = 0;
  if (p.paged) {
    if (p.kv_batch_idx != nullptr) cache_row = __builtin_amdgcn_readfirstlane(p.kv_batch_idx[b]);  // (uniform, and the
    if (p.leftpad_k != nullptr) leftpad = __builtin_amdgcn_readfirstlane(p.leftpad_k[b]);          // compiler must know)
    seqlen_k = seq_k[b] - leftpad;
    seqlen_k = seqlen_k > 0 ? seqlen_k : 0;
  } else {
    k_begin = seq_k[b];
    seqlen_k = seq_k[b + 1] - k_begin;
  }
  const int rows_total = seqlen_q * G;  // <= 16 row_groups
  // the host picked this kernel from the caller's max_seqlen_q: a sequence with more rows than that promise would be
  // written only in part. Fail the launch loudly instead of returning garbage rows.
  if (rows_total > kRowsPerWave * p.row_groups) __builtin_trap();
  const int r0 = rg * kRowsPerWave;  // first packed row of this workgroup
  if (rows_total <= r0) return;
  const int shift = seqlen_k - seqlen_q;

  const int my_row = r0 + l15;
  const bool row_ok = my_row < rows_total;
  const int my_qpos = row_ok ? my_row / G : 0;
  const int my_head = hk * G + (row_ok ? my_row % G : 0);
  const int q_abs = my_qpos + shift;
  const bool wave_rows_ok = r0 + kRowsPerWave <= rows_total;
  const int wave_qabs_lo = r0 / G + shift;
  const int wave_qabs_hi = ((r0 + kRowsPerWave <= rows_total ? r0 + kRowsPerWave : rows_total) - 1) / G + shift;

  int kv_hi = seqlen_k;
  if (p.causal_right >= 0) {
    const int lim = wave_qabs_hi + p.causal_right + 1;
    kv_hi = lim < kv_hi ? lim : kv_hi;
  }
  int kv_lo = 0;
  if (p.window_left >= 0) {
    const int lim = shift - p.window_left;
    kv_lo = lim > 0 ? lim : 0;
  }
  if (kv_hi < 0) kv_hi = 0;
  int t_lo = kv_lo / kTile, t_hi = (kv_hi + kTile - 1) / kTile;
  if (t_hi < t_lo) t_hi = t_lo;
  if (p.splits > 1) {
    const int per = (t_hi - t_lo + p.splits - 1) / p.splits;
    const int a = t_lo + split * per;
    const int e = a + per;
    t_lo = a < t_hi ? a : t_hi;
    t_hi = e < t_hi ? e : t_hi;
  }
  t_lo = __builtin_amdgcn_readfirstlane(t_lo);  // (uniform, and the compiler has to know: the tile loop is then a scalar loop)
  const int n_tiles = __builtin_amdgcn_readfirstlane(t_hi - t_lo);
  const int nw = n_tiles > wave ? (n_tiles - wave + NW - 1) / NW : 0;  // tiles of this wave: t_lo + wave + NW j

  const int pig = (0x2130 >> (4 * g4)) & 3;
  const int tau = (l15 & 3) | (((l15 >> 2) & 1) << 3) | (((l15 >> 3) & 1) << 2);

  v8s qf[KS];
  {
    const T* qrow = q + (int64_t)(q_begin + my_qpos) * p.q_s0 + (int64_t)my_head * p.q_s1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const v8s v = *reinterpret_cast<const v8s*>(qrow + 32 * ks + 8 * pig);
      const v8s zero = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = row_ok ? v : zero;
    }
  }

  // ---- addressing (the formula of the general kernel; a tile lies in one page, so one id per tile)
  const bool use_table = p.paged == 1;
  const int32_t* pg_src = use_table ? page_table + (int64_t)cache_row * p.table_stride : cu_q + b;
  const int pos_mask = use_table ? (1 << p.page_shift) - 1 : -1;
  const int pos_shift = use_table ? p.page_shift : 31;
  const int pos_base = p.paged ? leftpad : k_begin;
  const uint32_t kpg = use_table ? (uint32_t)p.k_s0 : 0u, vpg = use_table ? (uint32_t)p.v_s0 : 0u;
  const uint32_t kst = (uint32_t)(p.paged ? p.k_s1 : p.k_s0), vst = (uint32_t)(p.paged ? p.v_s1 : p.v_s0);
  const int64_t kbase_off = (p.paged == 2 ? (int64_t)cache_row * p.k_s0 + (int64_t)hk * p.k_s2
                                          : (int64_t)hk * (p.paged ? p.k_s2 : p.k_s1)) + 8 * pig;
  const int64_t vbase_row = (p.paged == 2 ? (int64_t)cache_row * p.v_s0 + (int64_t)hk * p.v_s2
                                          : (int64_t)hk * (p.paged ? p.v_s2 : p.v_s1));
  const int last_key = seqlen_k - 1;
  auto tile_of = [&](int j) { return t_lo + wave + NW * (j < nw ? j : nw - 1); };  // (past the end: the last tile again)
  // (a tile's two 16-token halves may lie in two pages - 16-token pages, round 5 late: the ids travel as one 64-bit value, low word
  //  = the first half's; from 32-token pages on both words are the same id)
  auto fetch_page = [&](int t) -> long long {
    int pos = t * kTile;
    pos = pos < last_key ? pos : last_key;
    const int a = pg_src[(pos + pos_base) >> pos_shift];
    int b2 = a;
    if (pos_shift < 5) {
      int pos2 = t * kTile + 16;
      pos2 = pos2 < last_key ? pos2 : last_key;
      b2 = pg_src[(pos2 + pos_base) >> pos_shift];
    }
    return (long long)(((unsigned long long)(uint32_t)b2 << 32) | (unsigned long long)(uint32_t)a);
  };
  // K fragment registers: [ks][half] token 16 half + tau(l15), dims 32 ks + 8 pig .. (16-bit: 16 bytes; fp8: 8 bytes,
  // widened right before the MFMA)
  using KV = typename std::conditional<KV8 == 0, v4i, v2i>::type;
  struct KRegs {
    KV k[2 * KS];
  };
  // V staging registers: 16 bytes per lane and load. 16-bit cache: chunk c of a token row = dims 8 c ..; a 128-dim block
  // row is 16 chunks (d = 64: 8). fp8: 16 bytes = 16 dims = two 16-byte chunks of the widened row.
  constexpr int CPR = D * ES / 16;               // 16-byte pieces per token row in the cache
  constexpr int TPL = 64 / CPR > 0 ? 64 / CPR : 1;  // tokens per load instruction (d = 256, 16-bit: 32 pieces -> 2 tokens)
  constexpr int NVL = kTile * CPR / 64;          // loads per tile and lane
  struct VRegs {
    v4i v[NVL];
  };
  const int v_piece = lane % CPR, v_tok = lane / CPR;  // this lane's piece of the row, its token within a load
  constexpr int CPRA = DA * ES / 16;                   // real pieces of a row; the positions past them re-fetch real ones
  const int v_src = v_piece < CPRA ? v_piece : v_piece - (CPR - CPRA);
  auto issue_k = [&](int t, long long page2, KRegs& r) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int pos = t * kTile + 16 * h + tau;
      pos = pos < last_key ? pos : last_key;
      const uint32_t cp = (uint32_t)((pos + pos_base) & pos_mask);
      const uint32_t page = h ? (uint32_t)((unsigned long long)page2 >> 32) : (uint32_t)page2;
      const int64_t off = (int64_t)((uint64_t)page * kpg + ((uint64_t)cp * kst + (uint64_t)kbase_off));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) r.k[2 * ks + h] = *reinterpret_cast<const KV*>(kcache + (off + 32 * ks) * ES);
    }
  };
  auto issue_v = [&](int t, long long page2, VRegs& r) {
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      int pos = t * kTile + TPL * i + v_tok;
      pos = pos < last_key ? pos : last_key;
      const uint32_t cp = (uint32_t)((pos + pos_base) & pos_mask);
      const uint32_t page = (TPL * i >= 16) ? (uint32_t)((unsigned long long)page2 >> 32) : (uint32_t)page2;  // (16 % TPL == 0)
      const int64_t off = (int64_t)((uint64_t)page * vpg + ((uint64_t)cp * vst + (uint64_t)vbase_row));
      r.v[i] = *reinterpret_cast<const v4i*>(vcache + off * ES + 16 * v_src);
    }
  };
  // fp8: 8 bytes -> 8 elements of T (exact), one v_cvt_scalef32_pk_* per pair, unit scale
  auto widen8 = [&](int lo, int hi) -> v4i {
    typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
    typedef _Float16 v2h_ __attribute__((ext_vector_type(2)));
    v4i r;
    const int x[2] = {lo, hi};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (std::is_same<T, bf16>::value && KV8 == 1) {
        r[2 * h] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x[h], 1.0f, true));
      } else if constexpr (std::is_same<T, bf16>::value) {
        r[2 * h] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(x[h], 1.0f, true));
      } else if constexpr (KV8 == 1) {
        r[2 * h] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x[h], 1.0f, true));
      } else {
        r[2 * h] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(x[h], 1.0f, false));
        r[2 * h + 1] = __builtin_bit_cast(int, (v2h_)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(x[h], 1.0f, true));
      }
    }
    return r;
  };
  auto kfrag = [&](const KV& x) -> v8s {
    if constexpr (KV8 == 0) return __builtin_bit_cast(v8s, x);
    else return __builtin_bit_cast(v8s, widen8(x[0], x[1]));
  };

  int vbase0;
  {
    const int qq = l15 >> 2, pp = l15 & 3;
    const int r = 8 * (g4 & 1) + 4 * (g4 >> 1) + qq;
    vbase0 = 256 * r + 16 * ((pp >> 1) ^ sw_main(r)) + 8 * (pp & 1);
  }
  char* vimg = smem + wave * (2 * VIMG);

  v4f o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float log2e = 1.4426950408889634f;
  const float kd = (KV8 != 0 && p.k_descale) ? p.k_descale[0] : 1.f, vd = (KV8 != 0 && p.v_descale) ? p.v_descale[0] : 1.f;
  const float scale = p.scale * kd, sc2 = scale * log2e;  // (the K descale multiplies every logit)
  __shared__ float xch_all[NW * 16];
  float* xch = xch_all + wave * 16;

  // Tile j of this wave: K(j) sits in kr (loaded two tiles ago), V(j) in vr (loaded one tile ago). Once V(j) is in its LDS
  // image and the scores are out of the matrix pipe, both register sets are free: V(j + 1) and K(j + 2) are requested
  // before the softmax. In flight per wave: K(j + 1), then V(j + 1) and K(j + 2).
  VRegs vr;
  auto compute = [&](int j, KRegs& kr, int buf, long long page_v, long long page_k) {
    const int t = tile_of(j);
    v4f s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s0 = M::run(kfrag(kr.k[2 * ks]), qf[ks], s0);
      s1 = M::run(kfrag(kr.k[2 * ks + 1]), qf[ks], s1);
    }
    // V image of this tile (wave-private: no barrier; LDS operations of a wave complete in order)
    char* vb = vimg + buf * VIMG;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int row = TPL * i + v_tok;
      if constexpr (KV8 == 0) {
        const int blk = v_piece >> 4, ch = v_piece & 15;  // 128-dim column block, 16-byte chunk inside it
        *reinterpret_cast<v4i*>(vb + blk * (kTile * 256) + row * 256 + ((ch ^ sw_main(row)) << 4)) = vr.v[i];
      } else {  // 16 fp8 -> two chunks of 8 elements
        // (the eight lanes of a ds_write_b128 group hold pieces 0..7 of ONE row: written "even chunk, then odd chunk" they hit
        //  chunks 0, 2, .., 14 - and chunks c, c + 8 share their 16 banks: SQ_LDS_BANK_CONFLICT 0.34 of the LDS cycles, round 4.
        //  The upper four lanes of a group - pieces 4..7, or at d = 64 the next token's pieces 0..3, whose swizzle key has the
        //  same parity - write their odd chunk first: every instruction covers all eight residues)
        const int blk = v_piece >> 3, ch = (v_piece & 7) * 2, first = (lane >> 2) & 1;
        char* rowp = vb + blk * (kTile * 256) + row * 256;
        const v4i w0 = widen8(vr.v[i][0], vr.v[i][1]), w1 = widen8(vr.v[i][2], vr.v[i][3]);
        *reinterpret_cast<v4i*>(rowp + (((ch + first) ^ sw_main(row)) << 4)) = first ? w1 : w0;
        *reinterpret_cast<v4i*>(rowp + (((ch + 1 - first) ^ sw_main(row)) << 4)) = first ? w0 : w1;
      }
    }
    issue_v(tile_of(j + 1), page_v, vr);
    issue_k(tile_of(j + 2), page_k, kr);
    const int tb = t * kTile + 8 * (g4 & 1) + 4 * (g4 >> 1);
    // (rows past the last one have q = 0: finite scores and weights, results never stored - no reason for the masked path)
    bool interior = p.softcap <= 0.f && (t * kTile + kTile <= seqlen_k);
    if (p.causal_right >= 0) interior = interior && (t * kTile + kTile - 1 <= wave_qabs_lo + p.causal_right);
    if (p.window_left >= 0) interior = interior && (t * kTile >= wave_qabs_hi - p.window_left);
    float m_new, m_use, alpha, psum = 0.f;
    v8s pf;
    if (interior) {
      float mt = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      m_new = fmaxf(m_run, mt * scale);
      m_use = m_new;
      alpha = __builtin_amdgcn_exp2f((m_run - m_use) * log2e);
      const float mneg = -m_use * log2e;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e], sc2, mneg));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e], sc2, mneg));
        psum += p0 + p1;
        pf[e] = M::cvt(p0);
        pf[4 + e] = M::cvt(p1);
      }
    } else {
      float z0[4], z1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = s0[e] * scale, c = s1[e] * scale;
        if (p.softcap > 0.f) {
          a = p.softcap * tanhf(a / p.softcap);
          c = p.softcap * tanhf(c / p.softcap);
        }
        const int k0 = tb + e, k1 = tb + 16 + e;
        bool m0 = !row_ok || k0 >= seqlen_k, m1 = !row_ok || k1 >= seqlen_k;
        if (p.causal_right >= 0) { m0 |= k0 > q_abs + p.causal_right; m1 |= k1 > q_abs + p.causal_right; }
        if (p.window_left >= 0) { m0 |= k0 < q_abs - p.window_left; m1 |= k1 < q_abs - p.window_left; }
        z0[e] = m0 ? -INFINITY : a;
        z1[e] = m1 ? -INFINITY : c;
      }
      float mt = fmaxf(fmaxf(fmaxf(z0[0], z0[1]), fmaxf(z0[2], z0[3])), fmaxf(fmaxf(z1[0], z1[1]), fmaxf(z1[2], z1[3])));
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      m_new = fmaxf(m_run, mt);
      m_use = m_new == -INFINITY ? 0.f : m_new;
      alpha = __builtin_amdgcn_exp2f((m_run - m_use) * log2e);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __builtin_amdgcn_exp2f((z0[e] - m_use) * log2e);
        const float p1 = __builtin_amdgcn_exp2f((z1[e] - m_use) * log2e);
        psum += p0 + p1;
        pf[e] = M::cvt(p0);
        pf[4 + e] = M::cvt(p1);
      }
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__any(alpha != 1.0f)) {
      if (lane < 16) xch[lane] = alpha;
      const v4f a4 = *reinterpret_cast<const v4f*>(xch + 4 * g4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        o[nt][0] *= a4[0]; o[nt][1] *= a4[1]; o[nt][2] *= a4[2]; o[nt][3] *= a4[3];
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const char* a = vb + (nt >> 3) * (kTile * 256) + (vbase0 ^ ((nt & 7) << 5));
      const v4s v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)SGLK_LDS(a));
      const v4s v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)SGLK_LDS(a + 4096));
      v8s vf;
      vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
      vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
      o[nt] = M::run(pf, vf, o[nt]);
    }
  };

  // ---- this wave's tiles. Page ids: p1 of tile j + 1 (for its V), p2 of tile j + 2 (for its K, then its V), p3 of tile
  // j + 3: fetched one step before they are first used, in front of the loads that follow.
  if (nw > 0) {
    KRegs ka, kb;
    long long p0 = fetch_page(tile_of(0)), p1 = fetch_page(tile_of(1)), p2 = fetch_page(tile_of(2));
    issue_k(tile_of(0), p0, ka);
    issue_v(tile_of(0), p0, vr);
    issue_k(tile_of(1), p1, kb);
    int j = 0;
    for (; j + 2 <= nw; j += 2) {
      long long p3 = fetch_page(tile_of(j + 3));
      compute(j, ka, 0, p1, p2);       // requests V(j + 1), K(j + 2)
      const long long p4 = fetch_page(tile_of(j + 4));
      compute(j + 1, kb, 1, p2, p3);   // requests V(j + 2), K(j + 3)
      p1 = p3;
      p2 = p4;
    }
    if (j < nw) compute(j, ka, 0, p1, p2);
  }

  // ---- merge the four waves' states into wave 0 (through the V images, everybody being done with them)
  __syncthreads();
  float* mo = reinterpret_cast<float*>(smem);            // [wave][nt][lane] v4f
  float* mm = mo + NW * NT * 64 * 4;                 // [wave][lane] m, then [wave][lane] l
  if (wave != 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<v4f*>(mo + ((wave * NT + nt) * 64 + lane) * 4) = o[nt];
    mm[wave * 64 + lane] = m_run;
    mm[(NW + wave) * 64 + lane] = l_run;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma nounroll
  for (int w = 1; w < NW; ++w) {
    const float m_w = mm[w * 64 + lane], l_w = mm[(NW + w) * 64 + lane];
    const float m_new = fmaxf(m_run, m_w);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float fa = __builtin_amdgcn_exp2f((m_run - m_use) * log2e), fb = __builtin_amdgcn_exp2f((m_w - m_use) * log2e);
    l_run = l_run * fa + l_w * fb;
    m_run = m_new;
    // row factors (this lane's row l15) -> the lanes holding output rows 4 g4 + e
    if (lane < 16) { xch[lane] = fa; xch_all[16 + lane] = fb; }
    const v4f a4 = *reinterpret_cast<const v4f*>(xch + 4 * g4), b4 = *reinterpret_cast<const v4f*>(xch_all + 16 + 4 * g4);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const v4f ow = *reinterpret_cast<const v4f*>(mo + ((w * NT + nt) * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[nt][e] = o[nt][e] * a4[e] + ow[e] * b4[e];
    }
  }

  // ---- epilogue (as the general kernel, rows of wave 0)
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const bool final_pass = p.splits == 1;
  const float m_fin = m_run;
  float lse_val = (l_tot > 0.f && m_fin != -INFINITY) ? m_fin + logf(l_tot) : -INFINITY;
  if (final_pass && p.sinks != nullptr && row_ok) {
    const float sk = p.sinks[my_head];
    const float m2 = fmaxf(m_fin, sk);
    const float l2 = l_tot * __builtin_amdgcn_exp2f((m_fin - m2) * log2e) + __builtin_amdgcn_exp2f((sk - m2) * log2e);
    lse_val = m2 + logf(l2);
    l_tot = (m_fin == -INFINITY) ? INFINITY : l_tot + __builtin_amdgcn_exp2f((sk - m_fin) * log2e);
  }
  const float inv_l = (l_tot > 0.f && l_tot < INFINITY) ? vd / l_tot : 0.f;  // (V descale folded in)
  if (lane < 16) xch[lane] = inv_l;
  const v4f i4 = *reinterpret_cast<const v4f*>(xch + 4 * g4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = r0 + 4 * g4 + e;
    if (row >= rows_total) continue;
    const int qpos = row / G, head = hk * G + row % G;
    const int64_t tok = q_begin + qpos;
    if (final_pass) {
      T* orow = (T*)p.out + tok * p.o_s0 + (int64_t)head * p.o_s1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) orow[nt * 16 + l15] = (T)(o[nt][e] * i4[e]);
    } else {
      float* orow = p.part_o + (((int64_t)split * p.total_q + tok) * p.Hq + head) * DA;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) orow[nt * 16 + l15] = o[nt][e] * i4[e];
    }
  }
  if (row_ok && g4 == 0) {
    const int64_t tok = q_begin + my_qpos;
    if (final_pass) p.lse[(int64_t)my_head * p.total_q + tok] = lse_val;
    else p.part_lse[((int64_t)split * p.Hq + my_head) * p.total_q + tok] = lse_val;
  }
}

// merge split-KV partials: out = sum_s exp(lse_s - L) O_s, L = log(sum_s exp(lse_s) [+ exp(sink)])
// One workgroup per (head, token). The launch sits behind 30 - 50 us decode kernels, so its own latency counts: the partial
// results of up to eight splits are requested TOGETHER with the log-sum-exps (one round trip instead of three dependent ones:
// maximum, denominator, weighted sum each re-read the log-sum-exps before) and the arithmetic runs on registers in split order.
template <typename T>
__global__ __launch_bounds__(128) void attn_reduce_kernel(T* __restrict__ out, float* __restrict__ lse_out,
                                                          const float* __restrict__ part_o,
                                                          const float* __restrict__ part_lse,
                                                          const float* __restrict__ sinks, int splits, int total_q,
                                                          int Hq, int D, int64_t o_s0, int64_t o_s1) {
  const int head = blockIdx.x;
  const int64_t tok = blockIdx.y;
  const float* pl = part_lse + (int64_t)head * total_q + tok;
  const int64_t ls = (int64_t)Hq * total_q;
  const float* po = part_o + (tok * Hq + head) * D;
  const int64_t os = (int64_t)total_q * Hq * D;
  const float sk = sinks ? sinks[head] : -INFINITY;
  if (splits <= 8 && D <= 256) {
    const int d0 = threadIdx.x, d1 = threadIdx.x + 128;
    float l[8], v0[8], v1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int sc = s < splits ? s : 0;  // (past the last split: split 0 again, weight 0)
      l[s] = pl[sc * ls];
      v0[s] = d0 < D ? po[sc * os + d0] : 0.f;
      v1[s] = d1 < D ? po[sc * os + d1] : 0.f;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < splits) mx = fmaxf(mx, l[s]);
    const float mref = fmaxf(mx, sk);
    float denom = 0.f;
    if (mref != -INFINITY) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < splits) denom += expf(l[s] - mref);
      if (sinks) denom += expf(sk - mref);
    }
    const float inv = denom > 0.f ? 1.0f / denom : 0.f;
    float a0 = 0.f, a1 = 0.f;
    if (mx != -INFINITY) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < splits && l[s] != -INFINITY) {
          const float w = expf(l[s] - mref);
          a0 += w * v0[s];
          a1 += w * v1[s];
        }
    }
    if (d0 < D) out[tok * o_s0 + (int64_t)head * o_s1 + d0] = (T)(a0 * inv);
    if (d1 < D) out[tok * o_s0 + (int64_t)head * o_s1 + d1] = (T)(a1 * inv);
    if (threadIdx.x == 0) lse_out[(int64_t)head * total_q + tok] = denom > 0.f ? mref + logf(denom) : -INFINITY;
    return;
  }
  float mx = -INFINITY;
  for (int s = 0; s < splits; ++s) mx = fmaxf(mx, pl[s * ls]);
  const float mref = fmaxf(mx, sk);
  float denom = 0.f;
  if (mref != -INFINITY) {
    for (int s = 0; s < splits; ++s) denom += expf(pl[s * ls] - mref);
    if (sinks) denom += expf(sk - mref);
  }
  const float inv = denom > 0.f ? 1.0f / denom : 0.f;
  for (int d = threadIdx.x; d < D; d += 128) {
    float acc = 0.f;
    if (mx != -INFINITY) {
      for (int s = 0; s < splits; ++s) {
        const float l = pl[s * ls];
        if (l != -INFINITY) acc += expf(l - mref) * po[s * os + d];
      }
    }
    out[tok * o_s0 + (int64_t)head * o_s1 + d] = (T)(acc * inv);
  }
  if (threadIdx.x == 0) lse_out[(int64_t)head * total_q + tok] = denom > 0.f ? mref + logf(denom) : -INFINITY;
}

template <typename T, int DKP, int KV8>
static int launch(hipStream_t st, const AttnParams& p, const void* q, const void* k, const void* v,
                  const int32_t* cu_q, const int32_t* seq_k, const int32_t* table, int batch, int max_rows) {
  constexpr int NB = (DKP * 2 + 255) / 256;
  constexpr int lds = (DKP > 256 ? 2 : 3) * 2 * NB * kTile * 256;
  static unsigned long long attr_done = 0;
  if (lds > 64 * 1024) {
    if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&attn_fwd_kernel<T, DKP, KV8>), lds, &attr_done, "fwd")) return rc;
  }
  dim3 grid((unsigned)cdiv(max_rows, kBlockM), (unsigned)(p.Hk * p.splits), (unsigned)batch);
  attn_fwd_kernel<T, DKP, KV8><<<grid, 256, lds, st>>>(p, (const T*)q, (const char*)k, (const char*)v, cu_q, seq_k, table);
  if (int rc = check_launch("fwd")) return rc;
  if (p.splits > 1) {
    attn_reduce_kernel<T><<<dim3(p.Hq, p.total_q), 128, 0, st>>>((T*)p.out, p.lse, p.part_o, p.part_lse, p.sinks,
                                                                  p.splits, p.total_q, p.Hq, p.D, p.o_s0, p.o_s1);
    return check_launch("fwd(reduce)");
  }
  return SGLK_OK;
}

template <typename T, int D, int NW, int MB, int KV8 = 0, int DA = D>
static int launch_prefill_nw(hipStream_t st, const AttnParams& p, const void* q, const void* k, const void* v,
                             const int32_t* cu_q, const int32_t* seq_k, const int32_t* table, int batch, int max_rows) {
  constexpr int lds = 2 * 2 * kPTile * D * 2;  // 64 KiB (d = 64: 32 KiB)
  static unsigned long long attr_done = 0;
  if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&attn_prefill_kernel<T, D, NW, MB, KV8, DA>), lds, &attr_done, "fwd")) return rc;
  dim3 grid((unsigned)cdiv(max_rows, 32 * MB * NW), (unsigned)(p.Hk * p.splits), (unsigned)batch);
  attn_prefill_kernel<T, D, NW, MB, KV8, DA><<<grid, 64 * NW, lds, st>>>(p, (const T*)q, (const char*)k, (const char*)v, cu_q, seq_k, table);
  if (int rc = check_launch("fwd(prefill)")) return rc;
  if (p.splits > 1) {
    attn_reduce_kernel<T><<<dim3(p.Hq, p.total_q), 128, 0, st>>>((T*)p.out, p.lse, p.part_o, p.part_lse, p.sinks,
                                                                  p.splits, p.total_q, p.Hq, p.D, p.o_s0, p.o_s1);
    return check_launch("fwd(reduce)");
  }
  return SGLK_OK;
}

#ifdef SGLK_PROBES
static int g_attn_prefill_waves = 0;  // 0: the policy below; 4 / 8: forced; 64: four waves of 64 rows (sglk_debug_set_attn_prefill_waves)
#else
constexpr int g_attn_prefill_waves = 0;
#endif

// Four waves (128 rows) per workgroup, two workgroups per CU: see the kernel comment.
template <typename T, int D, int KV8 = 0, int DA = D>
static int launch_prefill(hipStream_t st, const AttnParams& p, const void* q, const void* k, const void* v,
                          const int32_t* cu_q, const int32_t* seq_k, const int32_t* table, int batch, int max_rows) {
  if constexpr (DA != D) return launch_prefill_nw<T, D, 4, 1, KV8, DA>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
#ifdef SGLK_PROBES  // (the 8-wave form - one workgroup per CU - for A/B timing)
  if constexpr (KV8 == 0) {
    if (g_attn_prefill_waves == 8) return launch_prefill_nw<T, D, 8, 1>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
    if (g_attn_prefill_waves == 64) return launch_prefill_nw<T, D, 4, 2>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  }
#endif
  return launch_prefill_nw<T, D, 4, 1, KV8>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
}

template <typename T, int D, int KV8, int NW, int DA = D>
static int launch_decode_nw(hipStream_t st, const AttnParams& p, const void* q, const void* k, const void* v,
                            const int32_t* cu_q, const int32_t* seq_k, const int32_t* table, int batch) {
  constexpr int lds = NW * 2 * ((D + 127) / 128) * kTile * 256;  // four waves: 64 KiB (d = 256: 128 KiB); eight: 128 KiB
  static unsigned long long attr_done = 0;
  if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&attn_decode_kernel<T, D, KV8, NW, DA>), lds, &attr_done, "fwd")) return rc;
  dim3 grid((unsigned)(p.Hk * p.splits * p.row_groups), (unsigned)batch);
  attn_decode_kernel<T, D, KV8, NW, DA><<<grid, 64 * NW, lds, st>>>(p, (const T*)q, (const char*)k, (const char*)v, cu_q, seq_k, table);
  if (int rc = check_launch("fwd(decode)")) return rc;
  if (p.splits > 1) {
    attn_reduce_kernel<T><<<dim3(p.Hq, p.total_q), 128, 0, st>>>((T*)p.out, p.lse, p.part_o, p.part_lse, p.sinks,
                                                                  p.splits, p.total_q, p.Hq, p.D, p.o_s0, p.o_s1);
    return check_launch("fwd(reduce)");
  }
  return SGLK_OK;
}

#ifdef SGLK_PROBES
static int g_attn_decode_waves = 0;  // 0: the policy below; 4 / 8: forced (sglk_debug_set_attn_decode_waves)
static int g_attn_prefill_probe = 0;  // attn_prefill_kernel's timing probes (sglk_debug_set_attn_prefill_probe)
static unsigned long long* g_attn_prefill_stamps = nullptr;  // (sglk_debug_set_attn_prefill_stamps)
#else
constexpr int g_attn_decode_waves = 0;
constexpr int g_attn_prefill_probe = 0;
constexpr unsigned long long* g_attn_prefill_stamps = nullptr;
#endif

template <typename T, int D, int KV8>
static int launch_decode(hipStream_t st, const AttnParams& p, const void* q, const void* k, const void* v,
                         const int32_t* cu_q, const int32_t* seq_k, const int32_t* table, int batch) {
#ifdef SGLK_PROBES  // (eight waves, diagnostic build only: measured no faster, see the kernel comment; d = 256 needs the whole
  // register file with four)
  if constexpr (D <= 128) {
    if (g_attn_decode_waves == 8) return launch_decode_nw<T, D, KV8, 8>(st, p, q, k, v, cu_q, seq_k, table, batch);
  }
#endif
  return launch_decode_nw<T, D, KV8, 4>(st, p, q, k, v, cu_q, seq_k, table, batch);
}

// the decode kernel takes up to four 16-row groups of packed rows per (sequence, kv head)
constexpr int kDecodeRowsMax = 4 * kRowsPerWave;

template <typename T>
static int dispatch_dim(hipStream_t st, const AttnParams& p_in, const void* q, const void* k, const void* v,
                        const int32_t* cu_q, const int32_t* seq_k, const int32_t* table, int batch, int max_rows, int kv8) {
  AttnParams p = p_in;
  p.row_groups = max_rows <= kDecodeRowsMax ? (max_rows + kRowsPerWave - 1) / kRowsPerWave : 1;
  if (p.row_groups < 1) p.row_groups = 1;
  const int d = p.D;
  // prefill-sized problems at head dim 128 / 64 (Llama-3 / BASELINE configs[2]): the 128-row-block kernel. A row block must be
  // worth filling: at least 128 packed rows per (sequence, kv head) at the longest sequence.
  // (softcap - Gemma-2 - and d = 256 - Gemma; reference instantiation FMHAPrefillXe20.cmake:30-54 - run on this kernel since
  //  round 5; before, both fell through to the general 16-row kernel at ~0.08 - 0.11 of the bf16 peak. d = 256: 128 KiB of LDS and
  //  the whole register file, one workgroup per CU; 512-byte rows keep the d = 128 swizzle keys - (row & 15) for K, (row & 3) << 2
  //  for V - on the low bits of the 32 chunks of a row.)
  // (round 5, late: from 65 packed rows on - below, the decode kernel takes 16-row groups - and whatever num_splits says: the
  //  kernel does not split, it writes out and lse itself; 65 .. 127 rows and explicit split counts fell through to the general
  //  16-row kernel before - bs16 x 4096 keys at 31 query tokens 210 us, at 32 tokens 101 us)
  const bool prefill_rows = max_rows > kDecodeRowsMax;
  if (kv8 == 0 && (d == 128 || d == 64 || d == 256) && prefill_rows && p.q_s0 % 8 == 0 &&
      p.o_s0 % 4 == 0 && p.o_s1 % 4 == 0)
    return d == 128  ? launch_prefill<T, 128>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows)
           : d == 64 ? launch_prefill<T, 64>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows)
                     : launch_prefill<T, 256>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  // (head dims 96 / 192 - the reference builds both, FMHAPrefillXe20.cmake:30-54 - inside the 128 / 256 images since round 5; the
  //  DMA fetches whole 16-byte chunks of 192- / 384-byte rows: row strides and bases must be multiples of 8 elements)
  if (kv8 == 0 && (d == 96 || d == 192) && prefill_rows && p.q_s0 % 8 == 0 && p.q_s1 % 8 == 0 &&
      p.o_s0 % 4 == 0 && p.o_s1 % 4 == 0 && p.k_s0 % 8 == 0 && p.k_s1 % 8 == 0 && p.k_s2 % 8 == 0 && p.v_s0 % 8 == 0 &&
      p.v_s1 % 8 == 0 && p.v_s2 % 8 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)q % 16 == 0)
    return d == 96 ? launch_prefill<T, 128, 0, 96>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows)
                   : launch_prefill<T, 256, 0, 192>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  // (an fp8 cache at prefill sizes - the reference tests it, tests/test_flash_attention.py:1691-1704 - on the same kernel since
  //  round 5: registers instead of LDS-DMA, widened on the way into the 16-bit LDS images; rows are fetched in 16-byte pieces)
  if (kv8 != 0 && (d == 128 || d == 64) && prefill_rows && p.q_s0 % 8 == 0 && p.o_s0 % 4 == 0 &&
      p.o_s1 % 4 == 0 && p.k_s0 % 16 == 0 && p.k_s1 % 16 == 0 && p.k_s2 % 16 == 0 && p.v_s0 % 16 == 0 && p.v_s1 % 16 == 0 &&
      p.v_s2 % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0) {
    if (d == 128)
      return kv8 == 1 ? launch_prefill<T, 128, 1>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows)
                      : launch_prefill<T, 128, 2>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
    return kv8 == 1 ? launch_prefill<T, 64, 1>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows)
                    : launch_prefill<T, 64, 2>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  }
  // decode-sized problems at head dims 64 / 128 / 256 (16-bit or fp8 cache): every sequence has at most 16 packed rows per
  // kv head; a tile within one page
  if ((d == 64 || d == 128 || d == 256) && max_rows <= kDecodeRowsMax && (p.paged != 1 || p.page_shift >= 4) &&
      p.leftpad_k == nullptr && p.q_s0 % 8 == 0 &&
      (kv8 == 0 || (p.k_s0 % 16 == 0 && p.k_s1 % 16 == 0 && p.k_s2 % 16 == 0 && p.v_s0 % 16 == 0 && p.v_s1 % 16 == 0 &&
                    p.v_s2 % 16 == 0 && (uintptr_t)v % 16 == 0))) {  // (fp8: V rows are fetched in 16-byte pieces)
#define SGLK_DEC_GO(DD)                                                                                     \
  return kv8 == 0   ? launch_decode<T, DD, 0>(st, p, q, k, v, cu_q, seq_k, table, batch)                     \
         : kv8 == 1 ? launch_decode<T, DD, 1>(st, p, q, k, v, cu_q, seq_k, table, batch)                     \
                    : launch_decode<T, DD, 2>(st, p, q, k, v, cu_q, seq_k, table, batch)
    if (d == 64) SGLK_DEC_GO(64);
    if (d == 128) SGLK_DEC_GO(128);
    SGLK_DEC_GO(256);
#undef SGLK_DEC_GO
  }
  // (head dims 96 / 192 - the reference's paged decode builds them, FMHADecodeXe20.cmake:13-16 - inside the 128 / 256 forms of the
  //  same kernel since round 5; 16-byte loads of 192- / 384-byte rows: strides and bases in whole chunks)
  if ((d == 96 || d == 192) && max_rows <= kDecodeRowsMax && (p.paged != 1 || p.page_shift >= 4) && p.leftpad_k == nullptr &&
      p.q_s0 % 8 == 0 && p.q_s1 % 8 == 0 && (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 &&
      p.k_s0 % 16 == 0 && p.k_s1 % 16 == 0 && p.k_s2 % 16 == 0 && p.v_s0 % 16 == 0 && p.v_s1 % 16 == 0 && p.v_s2 % 16 == 0) {
#define SGLK_DEC_GO_A(DD, DA_)                                                                                  \
  return kv8 == 0   ? launch_decode_nw<T, DD, 0, 4, DA_>(st, p, q, k, v, cu_q, seq_k, table, batch)             \
         : kv8 == 1 ? launch_decode_nw<T, DD, 1, 4, DA_>(st, p, q, k, v, cu_q, seq_k, table, batch)             \
                    : launch_decode_nw<T, DD, 2, 4, DA_>(st, p, q, k, v, cu_q, seq_k, table, batch)
    if (d == 96) SGLK_DEC_GO_A(128, 96);
    SGLK_DEC_GO_A(256, 192);
#undef SGLK_DEC_GO_A
  }
  if (kv8 != 0) {  // fp8 KV cache: built for the head dims the reference exercises (and 64)
#define SGLK_FP8_GO(DKP)                                                                              \
  return kv8 == 1 ? launch<T, DKP, 1>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows)            \
                  : launch<T, DKP, 2>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows)
    if (d <= 64) SGLK_FP8_GO(64);
    if (d <= 128) SGLK_FP8_GO(128);
    if (d <= 256) SGLK_FP8_GO(256);
#undef SGLK_FP8_GO
    return fail(SGLK_EINVAL, "fwd: the fp8 KV cache path supports head dimensions up to 256, got %d", d);
  }
  if (d <= 64) return launch<T, 64, 0>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  if (d <= 96) return launch<T, 96, 0>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  if (d <= 128) return launch<T, 128, 0>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  if (d <= 192) return launch<T, 192, 0>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  if (d <= 256) return launch<T, 256, 0>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
  return launch<T, 512, 0>(st, p, q, k, v, cu_q, seq_k, table, batch, max_rows);
}

}  // namespace
}  // namespace sglk

// Split count used when the caller passes num_kv_splits == 0 ("auto"). Decode-sized problems (at most 16 packed rows per
// kv head: the four waves of a workgroup work on different tiles) want one workgroup per CU - measured at bs 16 x 8 kv
// heads x 4096 keys: 1 / 2 / 4 / 8 splits 53.8 / 53.5 / 55.5 / 60.3 us; otherwise ~2 workgroups per CU. At least 8 tiles
// (256 tokens) per split. Prefill-sized problems never split.
#ifdef SGLK_PROBES
extern "C" SGLK_API void sglk_debug_set_attn_decode_waves(int w) { sglk::g_attn_decode_waves = w; }
extern "C" SGLK_API void sglk_debug_set_attn_prefill_probe(int probe) { sglk::g_attn_prefill_probe = probe; }
extern "C" SGLK_API void sglk_debug_set_attn_prefill_waves(int w) { sglk::g_attn_prefill_waves = w; }
extern "C" SGLK_API void sglk_debug_set_attn_prefill_stamps(unsigned long long* buf) { sglk::g_attn_prefill_stamps = buf; }
#endif

extern "C" int64_t sglk_attn_auto_splits(int64_t batch, int64_t num_heads_k, int64_t max_rows_per_kv_head,
                                         int64_t max_seqlen_k) {
  // (up to 64 packed rows: the decode kernel, one workgroup per 16-row group)
  // (above: the 128-row-block kernel. This counted 64-row units until a sweep of explicit counts - lease zi - showed the rule at half
  //  the best count everywhere: bs 1 x 512 queries over 8192 / 32768 keys 118 / 360 us with its 2 splits, 93 / 281 with 4; bs 2 / 4 x 128
  //  over 32768 230 / 482 against 167 / 328; and splits shorter than 512 keys lose - 128 queries over 4096 keys 33 us with 8, 42 with 16)
  const int64_t wgs = batch * num_heads_k * (max_rows_per_kv_head <= 64 ? (max_rows_per_kv_head + 15) / 16 : (max_rows_per_kv_head + 127) / 128);
  const int64_t target = max_rows_per_kv_head <= 64 ? 256 : 512;
  if (wgs >= target * 3 / 4) return 1;
  const int64_t tiles = (max_seqlen_k + 31) / 32;
  int64_t s = target / (wgs > 0 ? wgs : 1);
  if (max_rows_per_kv_head <= 64) {
    // (lease zh, explicit split counts against this rule, bs x keys, us: 16 x 512 / 1024 / 2048 one split 15.6 / 21.2 / 30.8 against 17.2 /
    //  23.9 / 33.1 with two - half a round of workgroups is not worth a reduce launch below 128 tiles; 1 x 4096 eight splits 18.5 against
    //  22.5 / 23.1 with four / sixteen, 4 x 1024 and 1 x 1024 eight splits of 4 tiles 13.8 / 14.0 against 16 with four: splits of at least 4
    //  tiles, at most eight of them unless a split would then be longer than 64 tiles)
    if (wgs >= target / 2 && tiles < 128) return 1;
    const int64_t cap = tiles / 4, most = tiles / 64 > 8 ? tiles / 64 : 8;
    if (s > cap) s = cap;
    if (s > most) s = most;
  } else {
    // (one workgroup per CU over fewer than 16384 keys: a second one per CU through splits costs more than it brings - bs 16 x 64
    //  queries over 4096 keys 110 us unsplit, 122 with 2)
    if (wgs >= target / 2 && tiles < 512) return 1;
    const int64_t cap = tiles / 16;
    if (s > cap) s = cap;
  }
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

extern "C" int sglk_attn_fwd(sglk_stream_t stream, void* out, float* lse, const void* q, const void* k,
                             const void* v, const int32_t* cu_seqlens_q, const int32_t* seqlens_k,
                             const int32_t* page_table, const float* sinks, float* part_o, float* part_lse,
                             int64_t batch, int64_t total_q, int64_t max_seqlen_q, int64_t num_heads,
                             int64_t num_heads_k, int64_t head_dim, int64_t page_size, int64_t q_stride0,
                             int64_t q_stride1, int64_t o_stride0, int64_t o_stride1, int64_t k_stride0,
                             int64_t k_stride1, int64_t k_stride2, int64_t v_stride0, int64_t v_stride1,
                             int64_t v_stride2, int64_t table_stride, float softmax_scale, int is_causal,
                             int64_t window_left, int64_t window_right, float softcap, int64_t num_splits,
                             int dtype, int kv_dtype, const float* k_descale, const float* v_descale, int kv_layout,
                             const int32_t* kv_batch_idx, const int32_t* leftpad_k) {
  using namespace sglk;
  SGLK_REQUIRE(dtype == SGLK_BF16 || dtype == SGLK_F16, "mha_fwd only supports Half and BFloat16");
  SGLK_REQUIRE(num_heads > 0 && num_heads_k > 0 && num_heads % num_heads_k == 0,
               "Number of heads in key/value must divide number of heads in query");
  SGLK_REQUIRE(head_dim > 0 && head_dim <= 512, "FlashAttention forward only supports head dimension at most 512");
  SGLK_REQUIRE(head_dim % 8 == 0, "head_size should be a multiple of 8");
  const int kv8 = kv_dtype == SGLK_FP8_E4M3 ? 1 : kv_dtype == SGLK_FP8_E5M2 ? 2 : 0;
  SGLK_REQUIRE(kv8 != 0 || kv_dtype == dtype, "query and key must have the same dtype (or an fp8 e4m3 / e5m2 KV cache)");
  SGLK_REQUIRE(kv8 == 0 || (k_descale != nullptr && v_descale != nullptr), "fp8 KV cache requires k_descale and v_descale");
  SGLK_REQUIRE(q_stride0 % 8 == 0 && q_stride1 % 8 == 0 && k_stride0 % 8 == 0 && k_stride1 % 8 == 0 &&
                   k_stride2 % 8 == 0 && v_stride0 % 8 == 0 && v_stride1 % 8 == 0 && v_stride2 % 8 == 0 &&
                   (uintptr_t)q % 16 == 0 && (uintptr_t)k % (kv8 ? 8 : 16) == 0 && (uintptr_t)v % (kv8 ? 8 : 16) == 0,
               "fwd: q, k and v rows must be 16-byte aligned (8-byte for an fp8 cache)");
  SGLK_REQUIRE(kv_layout >= 0 && kv_layout <= 2, "fwd: kv_layout must be 0 (ragged), 1 (paged) or 2 (cache rows)");
  {
    const int64_t lim = (int64_t)1 << 31;
    SGLK_REQUIRE(k_stride0 >= 0 && k_stride1 >= 0 && k_stride2 >= 0 && v_stride0 >= 0 && v_stride1 >= 0 && v_stride2 >= 0 &&
                     k_stride0 < lim && k_stride1 < lim && k_stride2 < lim && v_stride0 < lim && v_stride1 < lim &&
                     v_stride2 < lim,
                 "fwd: k / v strides must be non-negative and below 2^31 elements");
  }
  SGLK_REQUIRE((kv_layout == 1) == (page_table != nullptr), "fwd: a page table goes with the paged layout and only with it");
  SGLK_REQUIRE(kv_layout != 0 || (kv_batch_idx == nullptr && leftpad_k == nullptr),
               "fwd: kv_batch_idx / leftpad_k need a KV cache (paged or cache-row layout)");
  const bool paged = kv_layout == 1;
  int page_shift = 0;
  if (paged) {
    SGLK_REQUIRE(page_size > 0 && (page_size & (page_size - 1)) == 0, "fwd: page size must be a power of two, got %lld",
                 (long long)page_size);
    while ((1ll << page_shift) < page_size) ++page_shift;
  }
  if (batch == 0 || total_q == 0) return SGLK_OK;
  SGLK_REQUIRE(max_seqlen_q >= 1 && max_seqlen_q <= total_q,
               "fwd: max_seqlen_q must be an upper bound of the query lengths in [1, total_q], got %lld",
               (long long)max_seqlen_q);
  SGLK_REQUIRE(num_splits >= 1, "fwd: num_splits must be resolved (>= 1) before the C-ABI call");
  SGLK_REQUIRE(num_splits == 1 || (part_o != nullptr && part_lse != nullptr), "fwd: split-KV needs partial buffers");
  AttnParams p;
  p.out = out;
  p.lse = lse;
  p.part_o = part_o;
  p.part_lse = part_lse;
  p.sinks = sinks;
  p.k_descale = kv8 ? k_descale : nullptr;
  p.v_descale = kv8 ? v_descale : nullptr;
  p.q_s0 = q_stride0; p.q_s1 = q_stride1;
  p.o_s0 = o_stride0; p.o_s1 = o_stride1;
  p.k_s0 = k_stride0; p.k_s1 = k_stride1; p.k_s2 = k_stride2;
  p.v_s0 = v_stride0; p.v_s1 = v_stride1; p.v_s2 = v_stride2;
  p.table_stride = table_stride;
  p.Hq = (int)num_heads;
  p.Hk = (int)num_heads_k;
  p.G = (int)(num_heads / num_heads_k);
  p.D = (int)head_dim;
  p.total_q = (int)total_q;
  p.page_shift = page_shift;
  p.paged = kv_layout;
  p.kv_batch_idx = kv_batch_idx;
  p.leftpad_k = leftpad_k;
  // causal == window_right 0 (reference flash_attention.cpp:401-404); negative = unlimited
  p.causal_right = is_causal ? 0 : (window_right >= 0 ? (int)window_right : -1);
  p.window_left = window_left >= 0 ? (int)window_left : -1;
  p.splits = (int)num_splits;
  p.scale = softmax_scale;
  p.softcap = softcap;
  p.probe = g_attn_prefill_probe;
  p.stamps = g_attn_prefill_stamps;
  const int max_rows = (int)(max_seqlen_q * p.G);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SGLK_BF16)
    return dispatch_dim<bf16>(st, p, q, k, v, cu_seqlens_q, seqlens_k, page_table, (int)batch, max_rows, kv8);
  return dispatch_dim<f16>(st, p, q, k, v, cu_seqlens_q, seqlens_k, page_table, (int)batch, max_rows, kv8);
}
