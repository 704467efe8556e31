// flash_mla_decode for gfx950: DeepSeek MLA decode over a paged latent KV cache.
//
// Replaces reference src/sycl/mla_decode.cpp:135-175 (host), :60-93 (split heuristic),
// :192-223 (workspace size) and src/sycl/kernels/mla/* (CuTe kernels). Contract kept
// (python/sgl_kernel/attention.py:54-132, tests/test_flash_mla_decode.py:38-59):
//   out[b,h,:512] = softmax(sm_scale * [q_nope[b,h], q_pe[b,h]] . cache[b,:,:576]^T) . cache[b,:,:512]
// over the first seq_lens[b] tokens of the pages page_table[b,:]; cache is [pages, PAGE, 576].
//
// Design (MI355X-first, the opposite of the reference's one-work-group-per-(batch,head) grid, which
// re-streams the whole latent cache once per head): ONE workgroup owns ALL heads of a batch element for
// a contiguous range of KV tokens, so each 1152-byte KV row is fetched from HBM once and is shared
// through LDS by the QK^T and the PV products.
//   - 8 waves (two per SIMD, 256 VGPRs each). Wave w owns heads 16w..16w+15: its Q^T fragments (72
//     VGPRs) and its fp32 O accumulator [16 heads x 512] (128 VGPRs) stay in registers for the whole
//     kernel. Waves beyond ceil(H/16) only help loading.
//   - KV tiles of 32 tokens go global->LDS by LDS-DMA (16 B/lane), 4-stage ring, three tiles in
//     flight behind a counted vmcnt, one raw s_barrier per tile.
//   - S^T = K . Q^T (16x16x32 MFMA, K rows read with ds_read_b128): the lane then owns one head and
//     8 of the tile's 32 tokens, so the online softmax is lane-local plus two cross-lane exchanges, and
//     the 16-bit P registers are directly the A operand of O += P . V, whose B operand (V = the same
//     LDS rows, first 512 columns) comes through ds_read_b64_tr_b16 (hardware transpose read).
//   - LDS image per tile: four [32 tokens][256 B] column blocks with the 16-byte chunk c of row r at
//     c ^ (((r&3)<<2)|((r>>2)&3)), plus a [32][128 B] block for the rope columns swizzled by (r>>1)&7.
//     Two free permutations make every LDS read conflict-free on that image: (1) within a 32-deep k
//     step, lane group g takes chunk pi(g), pi = (0,3,1,2), for BOTH MFMA operands; (2) MFMA row i
//     of a token tile is token tau(i) = i with bits 2 and 3 swapped, so that the two 4-row blocks a
//     32-lane half reads transposed are 8 rows apart.
//   - split-KV: grid = (splits, batch); partial O (normalised, fp32) and log2-sum-exp go to the
//     caller's workspace and a small second kernel merges them.
//   - More than 64 heads, and every prefill: mla_rows128_kernel further down (4 waves x 512 registers, two row tiles
//     per LDS fragment, O in fixed AGPRs); this kernel serves H <= 64 (and H > 64 through the W = 1 test hook).
//   - Fewer than 128 heads (template W = waves per 16-head group = 8, 4, 2 for H <= 16, 32, 64): the W waves of
//     a group share the work of every tile instead of idling. Each takes every W-th 32-deep k-step of QK^T and
//     leaves its partial S^T in LDS; after a second barrier every wave of the group adds the W partials in a
//     fixed order (all get the same bits), runs the same softmax, and accumulates only its 512/W output columns
//     of P.V. Per wave and tile that is 18/W + 32/W MFMA steps instead of 68, which keeps a 16-head decode on
//     the HBM roofline instead of on one wave's issue rate. (The partials take the CU's last 16 KiB of LDS next to the 4-stage ring.)
#include <math.h>

#include <type_traits>

#include "common.h"

namespace sglk {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

constexpr int kLatent = 512, kRope = 64, kQK = 576;
constexpr int kRowBytes = kQK * 2;     // 1152
constexpr int kTile = 32;              // kv tokens per tile
constexpr int kMainBytes = kTile * 1024;
constexpr int kRopeBytes = kTile * 128;
constexpr int kStageBytes = kMainBytes + kRopeBytes;  // 36 KiB
constexpr int kThreads = 512;
template <int W>
struct Cfg {
  static constexpr int kStages = 4;
  static constexpr int kPartOff = kStages * kStageBytes;  // W > 1: 8 waves x [2 token tiles][64 lanes][16 B]
  static constexpr int kLdsBytes = kPartOff + (W == 1 ? 0 : 8 * 2048);  // W > 1: exactly the CU's 160 KiB
  static constexpr int kNT = 32 / W;                       // 16-column output tiles per wave
  static constexpr int kKS = (18 + W - 1) / W;             // QK^T k-steps per wave
};

#define SGLK_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define SGLK_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

template <typename T>
struct Mfma;
template <>
struct Mfma<bf16> {
  static __device__ __forceinline__ v4f run(const v8s& a, const v8s& b, const v4f& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float x) { return __builtin_bit_cast(short, (bf16)x); }
  static __device__ __forceinline__ float back(short x) { return (float)__builtin_bit_cast(bf16, x); }
  // inline-asm forms with the accumulator pinned to the VGPR / AGPR file (rows128 kernel)
  static __device__ __forceinline__ void acc_v(v4f& c, const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
  template <int R>  // accumulator = the fixed registers a[R : R+3]
  static __device__ __forceinline__ void acc_agpr(const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(R), "i"(R + 3));
  }
};
template <>
struct Mfma<f16> {
  static __device__ __forceinline__ v4f run(const v8s& a, const v8s& b, const v4f& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float x) { return __builtin_bit_cast(short, (f16)x); }
  static __device__ __forceinline__ float back(short x) { return (float)__builtin_bit_cast(f16, x); }
  static __device__ __forceinline__ void acc_v(v4f& c, const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
  template <int R>  // accumulator = the fixed registers a[R : R+3]
  static __device__ __forceinline__ void acc_agpr(const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(R), "i"(R + 3));
  }
};

// ---- hand-managed accumulators in fixed AGPRs (rows128 kernel): the compiler never sees these values
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int R>
__device__ __forceinline__ void agpr_zero4() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\tv_accvgpr_write_b32 a[%2], 0\n\t"
               "v_accvgpr_write_b32 a[%3], 0" ::"i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R>
__device__ __forceinline__ void agpr_scale4(const v4f& f) {
  float t0, t1, t2, t3;
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%9]\n\tv_accvgpr_read_b32 %2, a[%10]\n\t"
      "v_accvgpr_read_b32 %3, a[%11]\n\ts_nop 1\n\t"
      "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %5\n\tv_mul_f32 %2, %2, %6\n\tv_mul_f32 %3, %3, %7\n\ts_nop 1\n\t"
      "v_accvgpr_write_b32 a[%8], %0\n\tv_accvgpr_write_b32 a[%9], %1\n\tv_accvgpr_write_b32 a[%10], %2\n\t"
      "v_accvgpr_write_b32 a[%11], %3"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
// two accumulator tuples (a[R .. R+7], the same four row factors) per statement: 8 reads, 8 multiplies, 8 writes, so
// that no instruction waits on the one before it (the 4-wide version was a serial chain: ~150 cycles per tuple)
template <int R>
__device__ __forceinline__ void agpr_scale8(const v4f& f) {
  float t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%12]\n\tv_accvgpr_read_b32 %1, a[%13]\n\tv_accvgpr_read_b32 %2, a[%14]\n\t"
      "v_accvgpr_read_b32 %3, a[%15]\n\tv_accvgpr_read_b32 %4, a[%16]\n\tv_accvgpr_read_b32 %5, a[%17]\n\t"
      "v_accvgpr_read_b32 %6, a[%18]\n\tv_accvgpr_read_b32 %7, a[%19]\n\t"
      "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %9\n\tv_mul_f32 %2, %2, %10\n\tv_mul_f32 %3, %3, %11\n\t"
      "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %9\n\tv_mul_f32 %6, %6, %10\n\tv_mul_f32 %7, %7, %11\n\t"
      "v_accvgpr_write_b32 a[%12], %0\n\tv_accvgpr_write_b32 a[%13], %1\n\tv_accvgpr_write_b32 a[%14], %2\n\t"
      "v_accvgpr_write_b32 a[%15], %3\n\tv_accvgpr_write_b32 a[%16], %4\n\tv_accvgpr_write_b32 a[%17], %5\n\t"
      "v_accvgpr_write_b32 a[%18], %6\n\tv_accvgpr_write_b32 a[%19], %7"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
      : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3), "i"(R + 4), "i"(R + 5),
        "i"(R + 6), "i"(R + 7));
}
template <int R>
__device__ __forceinline__ v4f agpr_read4() {
  float t0, t1, t2, t3;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
               "v_accvgpr_read_b32 %3, a[%7]\n\ts_nop 1"
               : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
               : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
  return (v4f){t0, t1, t2, t3};
}

// a[R .. R+7] *= f * f (one scalar factor, applied as two multiplies: the rescale factor 2^-(move) of the rows128z kernel
// spans up to ~2^-220, which a single fp32 factor cannot hold)
template <int R>
__device__ __forceinline__ void agpr_scale8_sq(float f) {
  float t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%9]\n\tv_accvgpr_read_b32 %1, a[%10]\n\tv_accvgpr_read_b32 %2, a[%11]\n\t"
      "v_accvgpr_read_b32 %3, a[%12]\n\tv_accvgpr_read_b32 %4, a[%13]\n\tv_accvgpr_read_b32 %5, a[%14]\n\t"
      "v_accvgpr_read_b32 %6, a[%15]\n\tv_accvgpr_read_b32 %7, a[%16]\n\t"
      "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
      "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8\n\t"
      "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
      "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8\n\t"
      "v_accvgpr_write_b32 a[%9], %0\n\tv_accvgpr_write_b32 a[%10], %1\n\tv_accvgpr_write_b32 a[%11], %2\n\t"
      "v_accvgpr_write_b32 a[%12], %3\n\tv_accvgpr_write_b32 a[%13], %4\n\tv_accvgpr_write_b32 a[%14], %5\n\t"
      "v_accvgpr_write_b32 a[%15], %6\n\tv_accvgpr_write_b32 a[%16], %7"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
      : "v"(f), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3), "i"(R + 4), "i"(R + 5), "i"(R + 6), "i"(R + 7));
}

__device__ __forceinline__ int sw_main(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
}

// hand-counted LDS wait that orders the asm reads of (lo, hi) in front of their first use
template <int N>
__device__ __forceinline__ void wait_lgkm(v2i& lo, v2i& hi) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit field");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "i"(N));
}

struct MlaParams {
  void* out;                 // [B, H, 512] T
  float* ws_o;               // [B, splits, H, 512] fp32 (splits > 1)
  float* ws_lse;             // [B, splits, H]
  unsigned long long* ws_cnt;  // [B, 2] {tickets taken, partial results published}: tagged words, see mla_cnt_up
  int64_t qn_sb, qn_sh, qp_sb, qp_sh;
  int64_t page_stride_bytes;
  int64_t table_stride;
  int H, page_shift, splits;
  // prefill (cu_seqlens_q given): rows of a workgroup are (token, head) pairs, 128 / 2^hp_shift tokens x 2^hp_shift
  // head slots; decode: hp_shift = 7 (one token, row = head)
  int hp_shift, causal;
  int probe;  // timing probe: 1 = stream the cache through LDS, compute nothing (garbage results)
  int q16;    // host only, rows128z kernel: QK^T on the 16-wide MFMA shape (launches that fill the chip for long: see S4)
  float scale_log2;
};

// q_nope [B,H,512] / q_pe [B,H,64] (strides in p), cache [pages, PAGE, 576], seq_lens [B], page_table
// [B, table_stride]: read-only for the whole launch (__restrict__ lets the page lookups be scalar loads,
// which keeps them off the vmcnt counter the LDS-DMA ring is timed with).
template <typename T, int W>
__global__ __launch_bounds__(kThreads, 2) void mla_decode_kernel(MlaParams p, const T* __restrict__ q_nope,
                                                                 const T* __restrict__ q_pe,
                                                                 const char* __restrict__ cache,
                                                                 const int32_t* __restrict__ seq_lens,
                                                                 const int32_t* __restrict__ page_table,
                                                                 const int32_t* __restrict__ cu_seqlens_q) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using M = Mfma<T>;
  using C = Cfg<W>;
  constexpr int kStages = C::kStages, kNT = C::kNT, kKS = C::kKS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hg = wave / W, ww = wave % W;  // head group, position inside the group
  const int split = blockIdx.x, b = blockIdx.y;
  const int H = p.H;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- rows of this workgroup. Decode: one token (the batch element), row = head. Prefill: tokens t0 .. t0+TPW-1
  // of sequence b, row = (token - t0) * Hp + head; every token sees the first kv_len(token) cache rows (causal:
  // bottom-right aligned, reference tests/test_flash_mla_prefill.py:73-81), the tile loop runs to the last one's.
  const int hp_mask = (1 << p.hp_shift) - 1;
  const int grp_tok = (hg * 16) >> p.hp_shift, grp_head0 = (hg * 16) & hp_mask;  // uniform per wave
  int q_row0 = b, n_tok = 1, seq, kv_first;  // first q row, valid tokens here, kv length of the last / first token
  if (cu_seqlens_q != nullptr) {
    const int q0 = cu_seqlens_q[b], sq = cu_seqlens_q[b + 1] - q0, sk = seq_lens[b];
    const int t0 = (int)blockIdx.z << (7 - p.hp_shift);
    if (t0 >= sq) return;
    const int tpw = 1 << (7 - p.hp_shift);
    n_tok = (sq - t0) < tpw ? (sq - t0) : tpw;
    q_row0 = q0 + t0;
    kv_first = p.causal ? sk - sq + t0 + 1 : sk;
    seq = p.causal ? sk - sq + t0 + n_tok : sk;
  } else {
    seq = seq_lens[b];
    kv_first = seq;
  }
  const bool active = grp_tok < n_tok && grp_head0 < H && p.probe != 1;
  // kv length of the row this lane scores (head slot l15 of the group); rows that are not stored behave like the
  // last token's so that their softmax stays finite
  const int kv_row = (cu_seqlens_q != nullptr && p.causal && grp_tok < n_tok) ? kv_first + grp_tok : seq;
  const int ntiles = (seq + kTile - 1) / kTile;
  const int tps = (ntiles + p.splits - 1) / p.splits;
  const int t_begin = split * tps;
  const int t_end = (t_begin + tps < ntiles) ? (t_begin + tps) : ntiles;

  const int32_t* table = page_table + (int64_t)b * p.table_stride;
  const int page_mask = (1 << p.page_shift) - 1;

  // ---- LDS-DMA of one tile. Main part: 4 column blocks x 8 row groups of [4 rows][256 B]; wave w
  // fills block w>>1, row groups 4(w&1)..+3. Rope part: waves 0..3 fill rows 8w..8w+7 of [32][128 B].
  auto stage_tile = [&](int t, int st) {
    char* base = smem + st * kStageBytes;
    const int tok0 = t * kTile;
    // a 32-token tile touches one page (PAGE >= 32) or two (PAGE == 16); if the second half of the tile is
    // past the sequence its page entry may be unused: re-use the first page (those rows are masked)
    const int pg0 = table[tok0 >> p.page_shift];
    int pg1 = pg0;
    if (p.page_shift == 4 && tok0 + 16 < seq) pg1 = table[(tok0 + 16) >> 4];
    const char* src0 = cache + (int64_t)pg0 * p.page_stride_bytes;
    const char* src1 = cache + (int64_t)pg1 * p.page_stride_bytes;
    const int cb = wave >> 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rg = (wave & 1) * 4 + i;
      const int row = rg * 4 + (lane >> 4);
      const int ch = (lane & 15) ^ sw_main(row);
      const int in_page = (tok0 + row) & page_mask;
      const char* src = ((wave & 1) == 0 ? src0 : src1) + in_page * kRowBytes + cb * 256 + ch * 16;
      __builtin_amdgcn_global_load_lds(SGLK_GLB(src), SGLK_LDS(base + cb * 8192 + rg * 1024), 16, 0, 0);
    }
    if (wave < 4) {
      const int row = wave * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      const int in_page = (tok0 + row) & page_mask;
      const char* src = (wave < 2 ? src0 : src1) + in_page * kRowBytes + 1024 + ch * 16;
      __builtin_amdgcn_global_load_lds(SGLK_GLB(src), SGLK_LDS(base + kMainBytes + wave * 1024), 16, 0, 0);
    }
  };

  if (t_begin >= t_end) {
    // nothing to do for this split (or an empty sequence)
    if (active && ww == 0) {
      for (int i = lane; i < 16 * kLatent; i += 64) {
        const int head = hg * 16 + i / kLatent, d = i % kLatent;
        if (head < H) {
          if (p.splits == 1) ((T*)p.out)[((int64_t)b * H + head) * kLatent + d] = (T)0.f;
          else p.ws_o[(((int64_t)b * p.splits + split) * H + head) * kLatent + d] = 0.f;
        }
      }
      if (p.splits > 1 && lane < 16 && hg * 16 + lane < H)
        p.ws_lse[((int64_t)b * p.splits + split) * H + hg * 16 + lane] = -INFINITY;
    }
    return;
  }

  // k permutation inside a 32-deep MFMA step: lane group g takes the 8 elements at 8*pi(g)
  const int pig = (0x2130 >> (4 * g)) & 3;  // pi = (0,3,1,2)

  // ---- Q^T fragments (MFMA B operand): lane (head l15, group g) holds q[head][32 ks + 8 pi(g) .. +7]
  // wave ww of a group takes k-steps ww, ww + W, ... (< 18; steps 16, 17 are the rope columns)
  v8s qf[kKS];
  {
    const int head = grp_head0 + l15;
    const bool ok = active && head < H;
    const int64_t qrow = q_row0 + (active ? grp_tok : 0);
    const T* qn = q_nope + qrow * p.qn_sb + (int64_t)(ok ? head : 0) * p.qn_sh + 8 * pig;
    const T* qp = q_pe + qrow * p.qp_sb + (int64_t)(ok ? head : 0) * p.qp_sh + 8 * pig;
    const v8s zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int jk = 0; jk < kKS; ++jk) {
      const int ks = ww + jk * W;
      const int kc = ks < 18 ? ks : 17;
      const T* src = kc < 16 ? qn + 32 * kc : qp + 32 * (kc - 16);
      const v8s v = *reinterpret_cast<const v8s*>(src);
      qf[jk] = (ok && ks < 18) ? v : zero;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Q is in registers before any LDS-DMA is counted

  // ---- per-lane LDS read offsets
  // K row read (MFMA A operand): MFMA row l15 of token tile tt is token 16 tt + tau(l15)
  const int tau = (l15 & 3) | (((l15 >> 2) & 1) << 3) | (((l15 >> 3) & 1) << 2);
  // (the second token tile is 16 rows = +4096 / +2048 bytes further: the swizzles only use row bits 0..3;
  //  stepping the chunk index by a power of two is an XOR on the byte offset, so one base per kind suffices)
  const int kbase = 256 * tau + 16 * (pig ^ sw_main(tau));                       // ^ (k4 << 6): chunk 4 k4 + pi(g)
  const int rbase = kMainBytes + 128 * tau + 16 * (pig ^ ((tau >> 1) & 7));      // ^ (k2 << 6)
  // transposed V read (MFMA B operand, 16 dims x 32 tokens): lane group g reads the 4-row blocks starting
  // at rows 8(g&1) + 4(g>>1) and +16; lane i = l15 supplies row q = i>>2, 8-byte piece pp = i&3 of the 32 bytes
  // of 16-dim tile n8 (chunks 2 n8, 2 n8 + 1) of a 128-column block:  vbase0 ^ (n8 << 5)
  int vbase0;
  {
    const int q = l15 >> 2, pp = l15 & 3;
    const int r = 8 * (g & 1) + 4 * (g >> 1) + q;
    vbase0 = 256 * r + 16 * ((pp >> 1) ^ sw_main(r)) + 8 * (pp & 1);
  }

  // W > 1: LDS offsets of this wave's k-steps (loop invariant) and of its first output tile
  int koff[kKS], kdelta[kKS];
#pragma unroll
  for (int jk = 0; jk < kKS; ++jk) {
    const int ks = ww + jk * W;
    const int kc = ks < 18 ? ks : 17;
    koff[jk] = kc < 16 ? (kc >> 2) * 8192 + (kbase ^ ((kc & 3) << 6)) : (rbase ^ ((kc - 16) << 6));
    kdelta[jk] = kc < 16 ? 4096 : 2048;  // second 16-token tile
  }
  const int g0 = ww * kNT;                                          // first output tile of this wave
  const int vxor_w = vbase0 ^ ((g0 & 7) << 5), vadd_w = (g0 >> 3) * 8192;

  v4f o[kNT];
#pragma unroll
  for (int nt = 0; nt < kNT; ++nt) o[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY;  // running max of the raw logits of head l15 (all four lane groups agree)
  float l_run = 0.f;        // running sum over this lane's own tokens
  // value held by the lane of head 4g + r (same lane group): per-head factors for the O rows this lane owns
  auto head_bcast = [&](float v, int r) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane & 48) | (4 * g + r)) << 2, __builtin_bit_cast(int, v)));
  };
  const uint32_t lds_base = (uint32_t)(uintptr_t)SGLK_LDS(smem);

  // ---- prologue: kStages - 1 tiles in flight
  const int n_my = t_end - t_begin;
#pragma unroll
  for (int i = 0; i < kStages - 1; ++i)
    if (i < n_my) stage_tile(t_begin + i, i);

  int st = 0;
  for (int i = 0; i < n_my; ++i) {
    const int t = t_begin + i;
    const int rem = n_my - 1 - i;  // tiles after this one (at most kStages - 2 of them are already in flight)
    if (wave < 4) {
      if (kStages == 4 && rem >= 2) wait_vmcnt<10>(); else if (rem >= 1) wait_vmcnt<5>(); else wait_vmcnt<0>();
    } else {
      if (kStages == 4 && rem >= 2) wait_vmcnt<8>(); else if (rem >= 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // tile t landed for every wave; every wave is done with tile t-1
    {
      const int st_next = st == 0 ? kStages - 1 : st - 1;  // the stage tile t-1 used
      if (i + kStages - 1 < n_my) stage_tile(t + kStages - 1, st_next);
    }
    const char* base = smem + st * kStageBytes;
    v4f s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (W == 1) {
      if (active) {
        // ---- S^T[token, head] = K . Q^T for the two 16-token tiles
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const v8s a0 = *reinterpret_cast<const v8s*>(base + (ks >> 2) * 8192 + (kbase ^ ((ks & 3) << 6)));
          const v8s a1 = *reinterpret_cast<const v8s*>(base + (ks >> 2) * 8192 + 4096 + (kbase ^ ((ks & 3) << 6)));
          s0 = M::run(a0, qf[ks], s0);
          s1 = M::run(a1, qf[ks], s1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const v8s a0 = *reinterpret_cast<const v8s*>(base + (rbase ^ (ks << 6)));
          const v8s a1 = *reinterpret_cast<const v8s*>(base + 2048 + (rbase ^ (ks << 6)));
          s0 = M::run(a0, qf[16 + ks], s0);
          s1 = M::run(a1, qf[16 + ks], s1);
        }
      }
    } else {
      // ---- this wave's share of S^T, left in LDS for the other waves of the group. The exchange uses inline asm
      // with hand-counted waits (plain LDS accesses next to LDS-DMA make the compiler drain vmcnt).
      const uint32_t part = lds_base + (uint32_t)C::kPartOff;
      if (active) {
        v4f p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jk = 0; jk < kKS; ++jk) {
          const v8s a0 = *reinterpret_cast<const v8s*>(base + koff[jk]);
          const v8s a1 = *reinterpret_cast<const v8s*>(base + koff[jk] + kdelta[jk]);
          p0 = M::run(a0, qf[jk], p0);
          p1 = M::run(a1, qf[jk], p1);
        }
        const uint32_t mine = part + (uint32_t)(wave * 2048 + lane * 16);
        // p0 / p1 come straight out of the matrix pipe: the compiler does not know that this asm reads them from
        // the LDS unit and inserts no wait states (seen: stale partials, run-to-run differences) - pad by hand
        asm volatile("s_nop 7\n\ts_nop 7\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" ::"v"(mine), "v"(p0),
                     "v"(p1)
                     : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (active) {
        const uint32_t grp = part + (uint32_t)(hg * W * 2048 + lane * 16);
        v4f q0[W], q1[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
          asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                       : "=&v"(q0[u]), "=&v"(q1[u])
                       : "v"(grp), "n"(u * 2048), "n"(u * 2048 + 1024)
                       : "memory");
        }
        // the wait names every register it releases, so that the additions cannot be scheduled above it
        if constexpr (W == 2) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0[0]), "+v"(q1[0]), "+v"(q0[1]), "+v"(q1[1])::"memory");
        } else if constexpr (W == 4) {
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(q0[0]), "+v"(q1[0]), "+v"(q0[1]), "+v"(q1[1]), "+v"(q0[2]), "+v"(q1[2]), "+v"(q0[3]), "+v"(q1[3])
                       :
                       : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(q0[0]), "+v"(q1[0]), "+v"(q0[1]), "+v"(q1[1]), "+v"(q0[2]), "+v"(q1[2]), "+v"(q0[3]), "+v"(q1[3]),
                         "+v"(q0[4]), "+v"(q1[4]), "+v"(q0[5]), "+v"(q1[5]), "+v"(q0[6]), "+v"(q1[6]), "+v"(q0[7]), "+v"(q1[7])
                       :
                       : "memory");
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
          s0 += q0[u];
          s1 += q1[u];
        }
      }
    }
    if (active) {
      // ---- online softmax for head l15 over this lane's 8 tokens (+ the other three lane groups')
      if (t * kTile + kTile > kv_first) {
        const int tb = t * kTile + 8 * (g & 1) + 4 * (g >> 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (tb + r >= kv_row) s0[r] = -INFINITY;
          if (tb + 16 + r >= kv_row) s1[r] = -INFINITY;
        }
      }
      float mt = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
      const float mneg = -m_new * p.scale_log2;
      float psum = 0.f;
      v8s pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], p.scale_log2, mneg));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], p.scale_log2, mneg));
        psum += p0 + p1;
        pf[r] = M::cvt(p0);
        pf[4 + r] = M::cvt(p1);
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
      // ---- rescale O when any head's maximum moved: O row (head) 4g + r needs alpha of head 4g + r
      if (__any(alpha != 1.0f)) {
        const v4f a4 = {head_bcast(alpha, 0), head_bcast(alpha, 1), head_bcast(alpha, 2), head_bcast(alpha, 3)};
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
          o[nt][0] *= a4[0]; o[nt][1] *= a4[1]; o[nt][2] *= a4[2]; o[nt][3] *= a4[3];
        }
      }
      // ---- O[head, dim] += P . V   (A = P from registers, B = V via transposed LDS reads).
      // (One read per asm statement, or early-clobber outputs: with two reads in one statement and plain "=v" the
      // compiler may give the first read's destination the address register, and the second read then races with
      // the first one's data return - seen as rare wrong tiles.)
      // The transposed reads are issued from inline asm: behind the builtin hipcc waits vmcnt(0) (it cannot
      // tell these LDS reads from the LDS-DMA writes in flight) and that would drain the prefetch ring every
      // tile. The 2 reads of one 16-dim tile are double buffered; LDS returns in order, so lgkmcnt(2) retires
      // the older pair. No other LDS/SMEM traffic of this wave may sit inside this section.
      __builtin_amdgcn_sched_barrier(0);
      {
        // this wave's kNT output tiles g0 .. g0 + kNT - 1 (W == 1: all 32); kDepth tile reads (2 instructions each)
        // stay in flight ahead of the MFMA that consumes them: one MFMA (16 cycles) does not cover an LDS round trip
        constexpr int kDepth = W == 1 ? 1 : 2, kNB = kDepth + 1;
        const uint32_t vb_t = lds_base + (uint32_t)(st * kStageBytes) + (uint32_t)vadd_w;
        v2i vb[kNB][2];
#define SGLK_TR_ISSUE(I)                                                                         \
  do {                                                                                           \
    const uint32_t a_ = vb_t + (uint32_t)(vxor_w ^ (((I) & 7) << 5));                            \
    /* one instruction per asm statement: the second destination may then reuse the address register, the */ \
    /* first one cannot (two reads in one statement with plain outputs raced: see the header)             */ \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                           \
                 : "=v"(vb[(I) % kNB][0]) : "v"(a_), "i"((((I) >> 3) * 8192)) : "memory");       \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                           \
                 : "=v"(vb[(I) % kNB][1]) : "v"(a_), "i"((((I) >> 3) * 8192 + 4096)) : "memory"); \
  } while (0)
#define SGLK_TR_WAIT(N, I) \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(vb[(I) % kNB][0]), "+v"(vb[(I) % kNB][1])::"memory")
#define SGLK_TR_STEP(G)                                                                                    \
  if constexpr ((G) < kNT) {                                                                               \
    if constexpr ((G) + kDepth < kNT) SGLK_TR_ISSUE((G) + kDepth);                                         \
    constexpr int ahead_ = (G) + kDepth < kNT ? kDepth : kNT - 1 - (G);                                    \
    if constexpr (ahead_ == 3) SGLK_TR_WAIT(6, G);                                                         \
    if constexpr (ahead_ == 2) SGLK_TR_WAIT(4, G);                                                         \
    if constexpr (ahead_ == 1) SGLK_TR_WAIT(2, G);                                                         \
    if constexpr (ahead_ == 0) SGLK_TR_WAIT(0, G);                                                         \
    v8s f_;                                                                                                \
    const v4s x0_ = __builtin_bit_cast(v4s, vb[(G) % kNB][0]), x1_ = __builtin_bit_cast(v4s, vb[(G) % kNB][1]); \
    f_[0] = x0_[0]; f_[1] = x0_[1]; f_[2] = x0_[2]; f_[3] = x0_[3];                                        \
    f_[4] = x1_[0]; f_[5] = x1_[1]; f_[6] = x1_[2]; f_[7] = x1_[3];                                        \
    o[G] = M::run(pf, f_, o[G]);                                                                           \
  }
        SGLK_TR_ISSUE(0);
        if constexpr (kDepth > 1) SGLK_TR_ISSUE(1);
        if constexpr (kDepth > 2) SGLK_TR_ISSUE(2);
        SGLK_TR_STEP(0) SGLK_TR_STEP(1) SGLK_TR_STEP(2) SGLK_TR_STEP(3) SGLK_TR_STEP(4) SGLK_TR_STEP(5)
        SGLK_TR_STEP(6) SGLK_TR_STEP(7) SGLK_TR_STEP(8) SGLK_TR_STEP(9) SGLK_TR_STEP(10) SGLK_TR_STEP(11)
        SGLK_TR_STEP(12) SGLK_TR_STEP(13) SGLK_TR_STEP(14) SGLK_TR_STEP(15) SGLK_TR_STEP(16) SGLK_TR_STEP(17)
        SGLK_TR_STEP(18) SGLK_TR_STEP(19) SGLK_TR_STEP(20) SGLK_TR_STEP(21) SGLK_TR_STEP(22) SGLK_TR_STEP(23)
        SGLK_TR_STEP(24) SGLK_TR_STEP(25) SGLK_TR_STEP(26) SGLK_TR_STEP(27) SGLK_TR_STEP(28) SGLK_TR_STEP(29)
        SGLK_TR_STEP(30) SGLK_TR_STEP(31)
#undef SGLK_TR_STEP
#undef SGLK_TR_WAIT
#undef SGLK_TR_ISSUE
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    st = st + 1 == kStages ? 0 : st + 1;
  }

  if (!active) return;
  // ---- epilogue: normalise by the row sums and write. O tile nt: lane holds dim 16 nt + l15, heads 4g + r
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv_l = 1.0f / l_tot;
  const v4f i4 = {head_bcast(inv_l, 0), head_bcast(inv_l, 1), head_bcast(inv_l, 2), head_bcast(inv_l, 3)};
  if (p.splits == 1) {
    T* out = (T*)p.out + (int64_t)(q_row0 + grp_tok) * H * kLatent;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int head = grp_head0 + 4 * g + r;
      if (head < H) {
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) out[(int64_t)head * kLatent + (g0 + nt) * 16 + l15] = (T)(o[nt][r] * i4[r]);
      }
    }
  } else {
    float* wo = p.ws_o + ((int64_t)b * p.splits + split) * H * kLatent;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int head = hg * 16 + 4 * g + r;
      if (head < H) {
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) wo[(int64_t)head * kLatent + (g0 + nt) * 16 + l15] = o[nt][r] * i4[r];
      }
    }
    if (ww == 0 && lane < 16 && hg * 16 + lane < H)
      p.ws_lse[((int64_t)b * p.splits + split) * H + hg * 16 + lane] = m_run * p.scale_log2 + log2f(l_tot);
  }
}


constexpr int kThreads2 = 256;

// ---------------------------------------------------------------------------------------------------------
// rows128 kernel, 32x32x16 form (round 3). Same contract, launch geometry, LDS image and DMA ring as the kernel above;
// what changes is the shape of the matrix instructions and with it who owns a row:
//   * S^T[32 tokens, 32 rows] = K . Q^T and O^T[32 dims, 32 rows] += V^T . P^T with v_mfma_f32_32x32x16: lane (l31, u)
//     owns ROW 32 wave + l31 in both products, so the running maximum, the row sum and the rescale factor are per-lane
//     scalars (one v_permlane32_swap joins the two token halves of a row) and the S accumulator, rounded to 16 bits, IS
//     the B operand of the second product (tokens in the order the accumulator holds them; the transposed V reads
//     fetch their tokens in that order). ~85 vector instructions per 32-token tile instead of 268.
//   * a 32-cycle MFMA holds the SIMD's vector issue port for 8 of its cycles (16x16x32: 8 of 16): the softmax of tile
//     j fits in the gaps of the P.V MFMAs of tile j-1 (<= 3 per gap, at most one v_exp), which the 16-cycle form could
//     not hide (measured on the kernel above: removing its softmax saved 0.49 us of a 2.85 us tile).
//   * the image "chunk c of row r at c ^ (((r&3)<<2)|((r>>2)&3))" serves the row reads (ds_read_b128: K as the A operand,
//     token l31, dims 16 ks + 8 u ..) and the transposed reads (ds_read_b64_tr_b16: V^T as the A operand, dims 32 dt +
//     l31, tokens 16 s + 8 e + 4 u ..) of this shape without bank conflicts and without the pi / tau permutations.
//   * registers: O^T 16 x 16 = 256 accumulators (the compiler puts them into the AGPR half), Q^T 36 x 4 = 144 VGPRs,
//     the rest rings and softmax state: all real MFMA builtins, so the compiler does the hazard and wait bookkeeping.
template <typename T>
struct Mfma32;
template <>
struct Mfma32<bf16> {
  static __device__ __forceinline__ v16f run(const v8s& a, const v8s& b, const v16f& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
  }
  template <int R>  // accumulator = the fixed registers a[R : R+15]
  static __device__ __forceinline__ void acc_agpr(const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(R), "i"(R + 15));
  }
  template <int R, int N>  // the same behind its own hand-counted LDS wait (one statement: no compiler nop in between)
  static __device__ __forceinline__ void wait_acc_agpr(const v8s& a, const v8s& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(R),
                 "i"(R + 15), "i"(N));
  }
  static __device__ __forceinline__ void acc_v(v16f& c, const v8s& a, const v8s& b) {  // accumulator pinned to VGPRs
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ void first_v(v16f& c, const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ int pack(float lo, float hi) {
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    const v2bf r = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(int, r);
  }
  static __device__ __forceinline__ float add2(int packed, float acc) {  // acc + lo + hi of the ROUNDED pair
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    const v2bf one = {(__bf16)1.0f, (__bf16)1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, packed), one, acc, false);
  }
};
template <>
struct Mfma32<f16> {
  static __device__ __forceinline__ v16f run(const v8s& a, const v8s& b, const v16f& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
  template <int R>
  static __device__ __forceinline__ void acc_agpr(const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(R), "i"(R + 15));
  }
  template <int R, int N>
  static __device__ __forceinline__ void wait_acc_agpr(const v8s& a, const v8s& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(R),
                 "i"(R + 15), "i"(N));
  }
  static __device__ __forceinline__ void acc_v(v16f& c, const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ void first_v(v16f& c, const v8s& a, const v8s& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ int pack(float lo, float hi) {
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    const v2h r = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(int, r);
  }
  static __device__ __forceinline__ float add2(int packed, float acc) {
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    const v2h one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h, packed), one, acc, false);
  }
};

#ifdef SGLK_PROBES
// in-kernel stamps (diagnostic build): per wave, shader cycles summed over the tiles of [0] the tile-landed wait, [1] the
// barrier, [2] the DMA issue, [3] QK^T, [4] P.V + softmax (+ rescale), [5] prologue, [6] epilogue, [7] tiles
__device__ unsigned long long g_mla_stamps[16 * 4 * 4096];
#endif
// kProbe (diagnostic build, garbage results): 1 = P.V without its LDS reads, 2 = QK^T without its LDS reads, 3 = both,
// 4 = no DMA, 5 = no reads and no DMA, 6 = no softmax micro-ops
// ---- split KV merged inside the kernel (rows128z kernel). The workgroup of a batch element that finishes LAST merges the
// splits itself: no second launch, and its own partial result never leaves the registers. Every workgroup takes a ticket
// when its tile loop is done; all but the last publish O / l (fp32, write-through stores) and their log2-sum-exp, then
// count themselves in; the last one waits until the others - which took their tickets before it, so they are running their
// epilogues - are counted in, and adds the partial results IN SPLIT ORDER (its own at its place): the sum does not depend
// on who came last. Hand-off as cdna_hip_programming.md guideline 16: sc1 payload stores, every storing wave drains,
// barrier, one agent-scope add; consumer: one lane polls relaxed, one agent acquire, barrier, loads.
//
// The two counters of a batch element need NO launch and no memset in front of the kernel (round 4 spent 4.8 us per call
// on a zeroing kernel; a memset node replayed from a captured graph did not even work). Each is a 64-bit word
// {48-bit tag, 16-bit count}, changed only by compare-and-swap: a word whose tag is not kCntTag - a fresh torch.empty
// workspace, or memory the allocator lent to somebody else in between - counts as zero, so the first arriver initialises
// it on the fly; the ticket word wraps to zero with the last ticket and the merger stores a zero count into the published
// word when it has seen everybody, so that the words a finished call leaves behind read zero for ANY split count of the
// next call. (A word of garbage that happens to carry the tag: 2^-48 per word. One workspace serves one call at a time.)
typedef __attribute__((address_space(1))) unsigned long long mla_gu64;
constexpr unsigned long long kCntTag = 0x5fa3c96d17b4ull;  // 48 bits
__device__ __forceinline__ uint32_t mla_cnt_of(unsigned long long w) { return (w >> 16) == kCntTag ? (uint32_t)(w & 0xffffu) : 0u; }
// count one up (wrapping to zero at `wrap`, 0 = never) and return the count found
__device__ __forceinline__ uint32_t mla_cnt_up(mla_gu64* w, uint32_t wrap) {
  unsigned long long old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    const uint32_t cnt = mla_cnt_of(old);
    uint32_t nxt = cnt + 1;
    if (wrap != 0u && nxt >= wrap) nxt = 0u;
    const unsigned long long want = (kCntTag << 16) | nxt;
    if (__hip_atomic_compare_exchange_strong(w, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return cnt;
  }
}
__device__ __forceinline__ bool mla_take_ticket(const MlaParams& p, int b, char* smem, int tid) {
  mla_gu64* cnt = (mla_gu64*)p.ws_cnt + 2 * (int64_t)b;
  uint32_t* lds_word = reinterpret_cast<uint32_t*>(smem);
  __builtin_amdgcn_s_barrier();  // (every wave has left the tile loop: the stages are free)
  if (tid == 0) *lds_word = mla_cnt_up(cnt, (uint32_t)p.splits);
  __syncthreads();
  return (int)(*lds_word) == p.splits - 1;
}
__device__ __forceinline__ void mla_count_in(const MlaParams& p, int b, int tid) {
  mla_gu64* cnt = (mla_gu64*)p.ws_cnt + 2 * (int64_t)b;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every storing wave, in front of the barrier)
  __builtin_amdgcn_s_barrier();
  if (tid == 0) mla_cnt_up(cnt + 1, 0u);
}
// false = the others did not show up within the bound (a lost workgroup, or a second call sharing this workspace): the
// caller then writes NaN instead of merging stale partial results
__device__ __forceinline__ bool mla_wait_others(const MlaParams& p, int b, char* smem, int tid) {
  mla_gu64* cnt = (mla_gu64*)p.ws_cnt + 2 * (int64_t)b;
  uint32_t* lds_word = reinterpret_cast<uint32_t*>(smem) + 1;  // (not the ticket's word: a slow wave may still be reading that)
  if (tid == 0) {
    // (bounded: a spin that never ends would take the device with it)
    uint32_t seen = 0u;
    for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
      seen = mla_cnt_of(__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if ((int)seen >= p.splits - 1) break;
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // everybody is counted in and nobody touches the word again in this call: leave a zero count for the next one
    __hip_atomic_store(cnt + 1, kCntTag << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_word = (int)seen >= p.splits - 1 ? 1u : 0u;
  }
  __syncthreads();
  return *lds_word != 0u;
}
// ---- epilogue of the rows128 kernel. A lane owns ROW 32 wave + l31 and, of that row, the dims 32 dt + 8 g + 4 u .. + 4 of
// register group i = 4 dt + g (64 groups of four accumulators). Stored as they lie, every instruction touches 32 rows - round
// 4's tail did, and was store-ISSUE-bound: 64 row-per-lane stores cost a wave ~37k cycles, the publishing and the merging
// workgroup of a batch element one after the other ~45 us of a 320-us launch (in-kernel stamps, round 5). Now
//   * partial results travel in FRAGMENT ORDER: group i of wave w, lane L at float4 index ((w 64 + i) 64 + L) of the split's
//     slab - one contiguous KiB per store / load instruction, the merging lane reads exactly what its twin wrote;
//   * the final rows go through LDS (the stages are free by then): a wave writes its 32 rows x 512 16-bit values row-major
//     (row stride 1040 B: the 16 lanes of a ds_write_b64 group hit 16 different bank pairs), reads them back a row per
//     instruction and stores one contiguous KiB per row.
constexpr int kOutOff = 1024;        // (the ticket words live in the first bytes)
constexpr int kOutRow = 1040;        // bytes per row in the LDS staging area
constexpr int kOutWave = 32 * kOutRow;
__device__ __forceinline__ int64_t mla_slab_f4(const MlaParams& p, int b, int split, int wave) {  // float4 index of a wave's slab
  return (((int64_t)b * p.splits + split) * 4 + wave) * 64 * 64;
}
// the 16-bit values of register group i of this lane's row -> the wave's staging area
template <typename M>
__device__ __forceinline__ void mla_stage_group(char* smem, int wave, int l31, int u, int i, const v4f& v) {
  const int lo = M::pack(v[0], v[1]), hi = M::pack(v[2], v[3]);
  *reinterpret_cast<v2i*>(smem + kOutOff + wave * kOutWave + l31 * kOutRow + (32 * (i >> 2) + 8 * (i & 3) + 4 * u) * 2) =
      (v2i){lo, hi};
}
// the wave's 32 staged rows -> out (row r of the workgroup = token slot r >> hp_shift, head r & hp_mask)
template <typename T>
__device__ __forceinline__ void mla_flush_rows(const MlaParams& p, char* smem, int wave, int lane, int q_row0, int n_tok) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int hp_mask = (1 << p.hp_shift) - 1;
  const char* src = smem + kOutOff + wave * kOutWave + lane * 16;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) {
    const int r = wave * 32 + k, tok = r >> p.hp_shift, head = r & hp_mask;
    const v4i v = *reinterpret_cast<const v4i*>(src + k * kOutRow);
    if (tok < n_tok && head < p.H)
      *reinterpret_cast<v4i*>((T*)p.out + ((int64_t)(q_row0 + tok) * p.H + head) * kLatent + lane * 8) = v;
  }
}
// The last workgroup's merge for the row of lane (l31, u): out = sum_s w_s O_s / sum_s w_s, w_s = 2^(lse_s - max lse), s in
// split order. OWN: this workgroup's O^T sits unnormalised in a0..a255 (x inv_l), log2-sum-exp my_lse; otherwise it had no
// keys. The merged rows are left in the wave's LDS staging area (mla_flush_rows stores them).
template <typename T, typename M, bool OWN>
__device__ __forceinline__ void mla_merge_splits(const MlaParams& p, int b, int split, int H, int hrow, char* smem, int wave,
                                                 int lane, float my_lse, float inv_l, bool all_in) {
  const int l31 = lane & 31, u = lane >> 5;
  const float* lse_b = p.ws_lse + (int64_t)b * p.splits * H + hrow;
  float mx = my_lse;
  for (int s2 = 0; s2 < p.splits; ++s2)
    if (s2 != split) mx = fmaxf(mx, lse_b[(int64_t)s2 * H]);
  float wsum = 0.f;
  for (int s2 = 0; s2 < p.splits; ++s2) {
    const float l2 = s2 == split ? my_lse : lse_b[(int64_t)s2 * H];
    if (l2 != -INFINITY) wsum += exp2f(l2 - mx);
  }
  // (all_in false: a partial result never arrived - NaN rows instead of a silently wrong sum)
  const float inv = !all_in ? __builtin_nanf("") : wsum > 0.f ? 1.0f / wsum : 0.f;
  static_for<0, 8>([&](auto cc) {  // 8 chunks of 8 register groups (32 accumulators of this lane's row)
    constexpr int c0 = decltype(cc)::value * 8;
    v4f acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < p.splits; ++s2) {
      const float l2 = s2 == split ? my_lse : lse_b[(int64_t)s2 * H];
      const float w = l2 == -INFINITY ? 0.f : exp2f(l2 - mx);
      if (w == 0.f) continue;
      if (s2 == split) {
        if constexpr (OWN) {
          static_for<0, 8>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const v4f v = agpr_read4<(c0 + k) * 4>();
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[k][e] += w * (v[e] * inv_l);
          });
        }
      } else {
        const v4f* src = reinterpret_cast<const v4f*>(p.ws_o) + mla_slab_f4(p, b, s2, wave) + (int64_t)c0 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const v4f v = src[k * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[k][e] += w * v[e];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const v4f o = {acc[k][0] * inv, acc[k][1] * inv, acc[k][2] * inv, acc[k][3] * inv};
      mla_stage_group<M>(smem, wave, l31, u, c0 + k, o);
    }
  });
}

// ---------------------------------------------------------------------------------------------------------
// rows128z kernel (round 5): round 3's 32x32x16 arithmetic, registers, LDS image and merge, with the tile loop re-cut around
// what round 5's in-kernel stamps showed (per 32-token tile, one wave per SIMD: 2176 cycles of MFMA, ~4000 spent):
//   * phase-start bubbles (~250 cycles each): every phase began with ~13 address instructions and the latency of its first
//     fragment reads. Now the P.V fragments of tile j-1 are requested in the last three MFMA gaps of QK^T(j) (K reads are asm
//     with hand-counted waits too, so the two rings share one counter), the 16 + 16 per-tile address registers are gone (one
//     stage-relative base per operand and a v_xor per chunk pair / dim-tile column, computed in the gaps that use them), and
//     the QK^T phase, which cannot start its reads before the tile barrier, fills that latency with the tile's rope DMA piece;
//   * LDS-DMA pieces (9 per wave and tile) cost ~60 cycles each in the QK^T gaps, ~240 cycles of barrier skew and a block of
//     64-bit vector address arithmetic: all four waves issued their piece in the same gap and queued for the CU's one
//     vector-memory path. Now the pieces sit in the P.V gaps (which carry no softmax exponentials), wave w in the gaps
//     m % 4 == w - at most one wave issues at a time - with scalar bases (`global_load_lds_dwordx4 voff, s[base]`) and
//     loop-invariant lane offsets;
//   * page ids: a 64-entry window of the page table lives in one VGPR (lane = entry), a tile's ids are two v_readlane; the
//     window is re-fetched (asm load, in front of the iteration's nine pieces, so that the next iteration's counted vmcnt
//     covers it) once per 32 - 128 tiles: no scalar load shares lgkmcnt with the LDS reads;
//   * softmax micro-ops re-cut so that no instruction directly follows its producer (fma, fma | exp, cvt | exp, dot2).
//
// kQ16 (round 5, second half): QK^T on v_mfma_f32_16x16x32 instead of 32x32x16. Same FLOP, same LDS bytes, twice the
// instructions - and the part holds a higher clock under them: a register-only stream of the 16-wide shape delivers
// 1.87 - 1.94 PFLOP/s against 1.64 - 1.70 for the 32-wide one on random bf16 operands, 1.47 - 1.59 against 1.37 - 1.48 with
// every operand re-read from LDS (tools/mfma_shape_probe.py; MI355X_MICROARCH.md, DVFS give-back item 7). The wave's 32 rows
// are two 16-row blocks rb, the tile's 32 tokens two 16-token blocks tb; a K fragment (16 tokens x 32 k, one ds_read_b128:
// MFMA row i = token 4 g(i >> 2) + (i & 3), g = (0, 2, 3, 1), which keeps the reads conflict-free on the unchanged image)
// feeds the two MFMAs of its row blocks. The accumulator of block (tb, rb) leaves lane (j, kg) with row 16 rb + j and the
// tokens 16 tb + 4 g(kg) + 0..3; eight v_permlane16_swap (odd 16-lane rows of the rb = 0 registers <-> even rows of the
// rb = 1 registers) hand every lane the 16 tokens of ROW lane & 31 - the ownership the softmax, P.V and the epilogue were
// written for. Only the order of a lane's tokens differs, which the transposed V reads follow (token half e swapped for
// the upper 32 lanes).
struct S4 {
  v4f q[4];  // before the swap: block (tb, rb) = q[2 tb + rb]; after: weights 8 tb + 4 h + r = q[2 tb + h][r]
};
template <int I>
__device__ __forceinline__ float sget(const v16f& s) { return s[I]; }
template <int I>
__device__ __forceinline__ void sset(v16f& s, float x) { s[I] = x; }
template <int I>
__device__ __forceinline__ float sget(const S4& s) { return s.q[I >> 2][I & 3]; }
template <int I>
__device__ __forceinline__ void sset(S4& s, float x) { s.q[I >> 2][I & 3] = x; }
__device__ __forceinline__ void s_pin(v16f& s) { asm volatile("" : "+v"(s)); }
__device__ __forceinline__ void s_pin(S4& s) { asm volatile("" : "+v"(s.q[0]), "+v"(s.q[1]), "+v"(s.q[2]), "+v"(s.q[3])); }
// (asm MFMA result -> VALU read wait states)
__device__ __forceinline__ void s_settle(v16f& s) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s)); }
__device__ __forceinline__ void s_settle(S4& s) {
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s.q[0]), "+v"(s.q[1]), "+v"(s.q[2]), "+v"(s.q[3]));
}
template <typename T, bool kStamp = false, int kProbe = 0, int kKA = 3, int kVA = 4, bool kQ16 = false>
__global__ __launch_bounds__(kThreads2, 1) void mla_rows128z_kernel(MlaParams p, const T* __restrict__ q_nope,
                                                                    const T* __restrict__ q_pe,
                                                                    const char* __restrict__ cache,
                                                                    const int32_t* __restrict__ seq_lens,
                                                                    const int32_t* __restrict__ page_table,
                                                                    const int32_t* __restrict__ cu_seqlens_q) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using M = Mfma32<T>;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, b = blockIdx.y;
  const int H = p.H;
  const int l31 = lane & 31, u = lane >> 5;

  // ---- this lane's row: (token slot, head) of workgroup row 32 wave + l31
  const int hp_mask = (1 << p.hp_shift) - 1;
  const int row = wave * 32 + l31;
  const int my_tok = row >> p.hp_shift, my_head = row & hp_mask;
  int q_row0 = b, n_tok = 1, seq, kv_first;
  if (cu_seqlens_q != nullptr) {
    const int q0 = cu_seqlens_q[b], sq = cu_seqlens_q[b + 1] - q0, sk = seq_lens[b];
    const int t0 = (int)blockIdx.z << (7 - p.hp_shift);
    if (t0 >= sq) return;
    const int tpw = 1 << (7 - p.hp_shift);
    n_tok = (sq - t0) < tpw ? (sq - t0) : tpw;
    q_row0 = q0 + t0;
    kv_first = p.causal ? sk - sq + t0 + 1 : sk;
    seq = p.causal ? sk - sq + t0 + n_tok : sk;
  } else {
    seq = seq_lens[b];
    kv_first = seq;
  }
  if (seq < 0) seq = 0;
  seq = __builtin_amdgcn_readfirstlane(seq);
  kv_first = __builtin_amdgcn_readfirstlane(kv_first);
  const bool ok = my_tok < n_tok && my_head < H;
  const bool work = __any(ok) && p.probe != 1;
  // the same, recomputed from a laundered thread id where it is needed after the prologue (the mask path, the epilogue): the
  // compiler otherwise carries these lane values across the tile loop - in registers it does not have (it parked them in
  // a0..a5, the accumulators this kernel owns: check_isa caught it)
  struct RowId {
    int u, my_tok, my_head;
    bool ok;
  };
  auto row_id = [&]() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    RowId r;
    const int ln = t & 63, r_row = wave * 32 + (ln & 31);
    r.u = ln >> 5;
    r.my_tok = r_row >> p.hp_shift;
    r.my_head = r_row & hp_mask;
    r.ok = r.my_tok < n_tok && r.my_head < H;
    return r;
  };
  const int ntiles = (seq + kTile - 1) / kTile;
  const int tps = (ntiles + p.splits - 1) / p.splits;
  int t_begin = split * tps;
  int t_end = (t_begin + tps < ntiles) ? (t_begin + tps) : ntiles;
#ifdef SGLK_PROBES
  // (probe 9: two splits of unequal length - the first one tile shorter: it finishes first, publishes while the second still
  //  streams, and the second, the merger, finds the partial result waiting)
  if (p.probe == 9 && p.splits == 2 && tps >= 16) {
    const int cut = tps - 1;
    t_begin = split ? cut : 0;
    t_end = split ? ntiles : cut;
  }
#endif
  const int n_my = t_end - t_begin;

  const int32_t* table = page_table + (int64_t)b * p.table_stride;
  const int page_mask = (1 << p.page_shift) - 1;

  // ---- nothing to do for this split (decode only: an empty sequence, or more splits than tiles)
  if (n_my <= 0) {
    if (ok) {
      if (p.splits == 1) {
        T* out = (T*)p.out + ((int64_t)(q_row0 + my_tok) * H + my_head) * kLatent;
        for (int d = u * 256; d < u * 256 + 256; ++d) out[d] = (T)0.f;
      }
    }
    if (p.splits > 1) {  // (no keys here: a log2-sum-exp of -inf takes part in the merge, no O - see mla_take_ticket)
      if (!mla_take_ticket(p, b, smem, tid)) {
        if (ok && u == 0)
          __hip_atomic_store(p.ws_lse + ((int64_t)b * p.splits + split) * H + my_head, -INFINITY, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        mla_count_in(p, b, tid);
      } else {
        const bool all_in = mla_wait_others(p, b, smem, tid);
        mla_merge_splits<T, M, false>(p, b, split, H, ok ? my_head : 0, smem, wave, lane, -INFINITY, 0.f, all_in);
        mla_flush_rows<T>(p, smem, wave, lane, q_row0, n_tok);
      }
    }
    return;
  }

  // ---- page-table window: lane L holds table[win0 + L] (entries past the sequence's last page: the last page)
  const int last_entry = (seq - 1) >> p.page_shift;
  auto tile_of = [&](int j) { return t_begin + (j < n_my ? j : n_my - 1); };
  int win0 = (tile_of(0) * kTile) >> p.page_shift;
  // (asm: the compiler must not count this load - it would drain the DMA ring in front of the first use - and the register
  //  is an IN-OUT operand of every statement that touches it: the load lands long after its statement, so the value must
  //  never be copied in between. `go` = 0 skips the load inside the statement: no branch merge, no phi, no copy)
  int v_ids = 0;
  auto ids_fetch = [&](int go) {
    int e = win0 + lane;
    e = e < last_entry ? e : last_entry;
    const uint32_t off = (uint32_t)e * 4u;
    asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\tglobal_load_dword %0, %1, %2\n1:"
                 : "+v"(v_ids)
                 : "v"(off), "s"(table), "s"(go)
                 : "memory", "scc");
  };
  ids_fetch(1);
  // page ids of tile t (both 16-token halves; only 16-token pages give them different ones). Scalar work is issue time of
  // the one wave a SIMD has: 32-bit index arithmetic, a 32 x 32 -> 64-bit product per page (the host checks the stride)
  const bool two_pages = p.page_shift == 4;
  const uint32_t stride32 = (uint32_t)p.page_stride_bytes;
  auto ids_of = [&](int t, int& pg0, int& pg1) {
    const int tok0 = t * kTile;
    const int i0 = (tok0 >> p.page_shift) - win0;
    pg0 = __builtin_amdgcn_readlane(v_ids, i0);
    pg1 = pg0;
    if (two_pages) pg1 = __builtin_amdgcn_readlane(v_ids, (tok0 + 16 < seq) ? i0 + 1 : i0);
  };
  // moves the window when tile t's entries do not lie in it (t: the tile whose ids the NEXT iteration reads)
  auto ids_advance = [&](int t) {
    const int e0 = (t * kTile) >> p.page_shift;
    const int go = __builtin_amdgcn_readfirstlane((e0 + 1 - win0 >= 64) ? 1 : 0);
    win0 = go ? e0 : win0;
    ids_fetch(go);
  };

  // ---- Q^T fragments (B operand of K . Q^T): lane supplies q[row][16 ks + 8 u .. + 8). The 32 fragments of the latent part
  // stay in registers (128); the four rope fragments sit in the 16 KiB of LDS the four stages leave free - each lane reads
  // back what it wrote, so no barrier - and come in through the K ring's counter: 16 registers the tile loop needs for the
  // P.V fragments it now requests while the QK^T phase still runs (with all of Q in registers the compiler parked ten
  // values in a0..a9, the accumulators this kernel owns).
  constexpr int kQpeOff = 4 * kStageBytes;
  v8s qf[32];
  const uint32_t qpe_addr = (uint32_t)(uintptr_t)SGLK_LDS(smem) + (uint32_t)(kQpeOff + wave * 4096 + lane * 16);
  if constexpr (kQ16) {
    // 16-wide shape: lane (j, kg) supplies q[row 16 rb + j][32 ks + 8 kg .. + 8) of both row blocks: qf[16 rb + ks] for the
    // latent k-steps, LDS slot 2 (ks - 16) + rb for the two rope steps
    const int j16 = lane & 15, kg = lane >> 4;
    const v8s zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int r2 = wave * 32 + 16 * rb + j16;
      const int tok2 = r2 >> p.hp_shift, head2 = r2 & hp_mask;
      const bool ok2 = tok2 < n_tok && head2 < H;
      const int64_t qrow = q_row0 + (ok2 ? tok2 : 0);
      const T* qn = q_nope + qrow * p.qn_sb + (int64_t)(ok2 ? head2 : 0) * p.qn_sh + 8 * kg;
      const T* qp = q_pe + qrow * p.qp_sb + (int64_t)(ok2 ? head2 : 0) * p.qp_sh + 8 * kg;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const v8s v = *reinterpret_cast<const v8s*>(qn + 32 * ks);
        qf[16 * rb + ks] = ok2 ? v : zero;
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const v8s v = *reinterpret_cast<const v8s*>(qp + 32 * k2);
        *reinterpret_cast<v8s*>(smem + kQpeOff + wave * 4096 + (2 * k2 + rb) * 1024 + lane * 16) = ok2 ? v : zero;
      }
    }
  } else {
    const int64_t qrow = q_row0 + (ok ? my_tok : 0);
    const T* qn = q_nope + qrow * p.qn_sb + (int64_t)(ok ? my_head : 0) * p.qn_sh + 8 * u;
    const T* qp = q_pe + qrow * p.qp_sb + (int64_t)(ok ? my_head : 0) * p.qp_sh + 8 * u;
    const v8s zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const v8s v = *reinterpret_cast<const v8s*>(qn + 16 * ks);
      qf[ks] = ok ? v : zero;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const v8s v = *reinterpret_cast<const v8s*>(qp + 16 * ks);
      *reinterpret_cast<v8s*>(smem + kQpeOff + wave * 4096 + ks * 1024 + lane * 16) = ok ? v : zero;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(v_ids)::"memory");  // Q and the page ids are in place before any LDS-DMA is counted

  // ---- LDS-DMA of one tile: wave w fills column block w (eight 1-KiB pieces of four rows each) and rope rows 8w .. 8w+7.
  // Global side: a scalar base (the tile's first row / its 17th) + a loop-invariant lane offset; the swizzle of the LDS
  // image sits in the lane offset (the DMA writes consecutive LDS addresses).
  const uint32_t dma_lo0 = (uint32_t)((lane >> 4) * kRowBytes + wave * 256 + 16 * ((lane & 15) ^ ((lane >> 4) << 2)));
  uint32_t dvo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dvo[r] = (dma_lo0 ^ (uint32_t)(r << 4)) + (uint32_t)(r * 4 * kRowBytes);
  const uint32_t dro = (uint32_t)((((wave & 1) * 8 + (lane >> 3)) * kRowBytes) + 1024 +
                                  16 * ((lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7)));
  const uint32_t lds_base = (uint32_t)(uintptr_t)SGLK_LDS(smem);
  // (the DMA instructions are asm text: an LDS-DMA the compiler can see makes it wait vmcnt(0) in front of the next LDS
  //  read it can see - any read may alias the DMA's LDS write - which would empty the ring once per tile)
  auto dma_s = [&](const char* sbase, uint32_t voff, uint32_t lds_dst) {
    if constexpr (kProbe == 7)  // (cache policy probe: non-temporal loads for a cache that is read once)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_dst)
                   : "memory", "m0");
    else
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
                   : "memory", "m0");
  };
  struct TileSrc {
    const char* sA;  // rows 0..15 of the tile
    const char* sB;  // rows 16..31
    uint32_t dst;    // this wave's column block in the tile's stage
    uint32_t rdst;   // this wave's rope rows
  };
  auto tile_src = [&](int t, int st, int pg0, int pg1) {
    TileSrc s;
    const int tok0 = t * kTile;
    s.sA = cache + ((uint64_t)(uint32_t)pg0 * stride32 + (uint32_t)((tok0 & page_mask) * kRowBytes));
    s.sB = two_pages ? cache + (uint64_t)(uint32_t)pg1 * stride32 : s.sA + 16 * kRowBytes;
    s.dst = lds_base + (uint32_t)(st * kStageBytes + wave * 8192);
    s.rdst = lds_base + (uint32_t)(st * kStageBytes + kMainBytes + wave * 1024);
    return s;
  };
  auto dma_piece = [&](const TileSrc& s, auto ic) {  // piece 0..7: a row group of the column block; 8: the rope rows
    constexpr int rg = decltype(ic)::value;
    if constexpr (kProbe != 4 && kProbe != 5) {
      if constexpr (rg < 8) dma_s(rg < 4 ? s.sA : s.sB, dvo[rg & 3], s.dst + (uint32_t)(rg * 1024));
      else dma_s(wave < 2 ? s.sA : s.sB, dro, s.rdst);
    }
  };
  auto stage_tile = [&](int t, int st, int pg0, int pg1) {
    const TileSrc s = tile_src(t, st, pg0, pg1);
    static_for<0, 9>([&](auto ic) { dma_piece(s, ic); });
  };

  // ---- per-lane LDS read offsets inside a stage
  // K row read, k-step ks: block ks / 8, chunk 2 (ks % 8) + u of token row l31  ->  kbase ^ (32 (ks % 8)) + 8192 (ks / 8)
  // kQ16: K fragment of k-step ks = 4 cb + ks', token block tb: lane (i, kg) reads chunk 4 ks' + kg of token 16 tb + tau(i),
  // tau(i) = 4 g(i >> 2) + (i & 3)  ->  kbase ^ (64 ks') + 8192 cb + 4096 tb; rope step k2: rbase ^ (64 k2) + 2048 tb
  const int i16k = lane & 15, kgk = lane >> 4, g16 = (0x78 >> (2 * (i16k >> 2))) & 3, tau16 = 4 * g16 + (i16k & 3);
  const uint32_t kbase = kQ16 ? (uint32_t)(256 * tau16 + 16 * (4 * (i16k & 3) + (kgk ^ g16)))
                              : (uint32_t)(256 * l31 + 16 * (u ^ sw_main(l31)));
  const uint32_t rbase = kQ16 ? (uint32_t)(kMainBytes + 128 * tau16 + 16 * (kgk ^ ((tau16 >> 1) & 7)))
                              : (uint32_t)(kMainBytes + 128 * l31 + 16 * (u ^ ((l31 >> 1) & 7)));  // rope: ^ (32 (ks - 32))
  // V^T transposed read of dim tile dt, k-step s, token half e (tokens 16 s + 8 e + 4 u + qq, dims 32 dt + 16 hh + 4 pp ..):
  //   8192 (dt / 4) + 4096 s + 2048 e + (vbase ^ (64 (dt % 4)) ^ (32 e))
  const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3, hh = (lane >> 4) & 1;
  const uint32_t vbase = (uint32_t)(256 * (4 * u + qq) + 8 * (pp & 1) + 16 * ((qq << 2) | ((2 * hh + (pp >> 1)) ^ u)));
  // token halves e = 0 / 1 of a k-step, relative to the stage. kQ16: behind the row swap a lane holds its tokens in the order
  // 16 tb + {0..3, 8..11} (lanes 0..31) / 16 tb + {12..15, 4..7} (lanes 32..63): the upper lanes fetch half 1 first
  const uint32_t vbase1 = (vbase ^ 32u) + 2048u;
  const uint32_t vbA = (kQ16 && u) ? vbase1 : vbase, vbB = (kQ16 && u) ? vbase : vbase1;

  // O^T tile dt = a[16 dt .. 16 dt + 15], named only in asm text; the clobbers tell the compiler that the kernel owns the
  // whole AGPR file (build.py check_isa verifies that it places nothing there)
  asm volatile("" ::: "a0", "a255");
  static_for<0, 64>([&](auto ic) { agpr_zero4<decltype(ic)::value * 4>(); });

  // Lazy reference (the idea of the 16-row kernel above, wider here): bf16 weights have fp32's exponent range, so the
  // reference need not be the maximum. When it moves it is set 2^kHead ABOVE the row's running maximum (weights restart at
  // 2^-kHead) and it moves again only when a weight would pass 2^kLazy: kHead + kLazy = 190 binades of growth per move (100
  // in the kernel above; the reference benchmark's q x 100 logits grow by ~220 binades over a split, and every move of any
  // of a wave's 32 rows costs the wave - and, through the tile barrier, the workgroup - a 768-instruction pass over O).
  // Range: 2^kLazy x keys x |V| stays below 2^127 for a million keys and |V| < 2^16; a weight 2^-24 below the largest
  // one is still >= 2^-124 (normal); the rescale factor 2^-(move), down to ~2^-220, is applied as two multiplies by its
  // square root. f16 weights must stay in [2^-14, 2^16): threshold 2^8, no headroom.
  constexpr bool kWide = std::is_same<T, bf16>::value;
  constexpr float kLazy = kWide ? 90.0f : 8.0f, kHead = kWide ? 100.0f : 0.0f;
  const float sl2 = p.scale_log2;
  const float head_raw = (kWide && sl2 > 0.f) ? kHead / sl2 : 0.f;  // 2^kHead in units of the raw logits
  float m_ref = -INFINITY, m_run = -INFINITY, l_run = 0.f;

  // Every tile slot past the split's last tile re-loads the last tile into a stage nobody reads (two duplicates per split):
  // the DMA needs no condition and every tile-landed wait is the same vmcnt(9).
  {
    int a0, a1;
    ids_of(tile_of(0), a0, a1);
    stage_tile(tile_of(0), 0, a0, a1);
    ids_of(tile_of(1), a0, a1);
    stage_tile(tile_of(1), 1, a0, a1);
  }

  float mt = 0.f, mb = 0.f, alpha = 1.0f, mneg = 0.f, psum = 0.f;
  bool upd = false;
  v8s pf[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};

  // second half of the softmax of the tile whose raw logits are in sp (reference mneg, fixed by the first half), as micro-ops
  // for the QK^T gaps 1 .. 27. Pair pr = weights 2 pr, 2 pr + 1: gap 3 pr + 1 both multiply-adds, gap 3 pr + 2 the first
  // exponential and the PREVIOUS pair's rounding, gap 3 pr + 3 the second exponential and the previous pair's row-sum step:
  // no instruction directly follows its producer (a dependent pair waits out the VALU latency: stamps, round 5).
  int pk_prev = 0;
  // the pieces: multiply-adds of pair pr, one exponential, rounding + packing of pair pq, its row-sum step, the tile's sum
  auto e_fma = [&](auto prc, auto& sp) {
    constexpr int pr = decltype(prc)::value;
    float y0 = __builtin_fmaf(sget<2 * pr>(sp), sl2, mneg), y1 = __builtin_fmaf(sget<2 * pr + 1>(sp), sl2, mneg);
    asm volatile("" : "+v"(y0), "+v"(y1));
    sset<2 * pr>(sp, y0);
    sset<2 * pr + 1>(sp, y1);
  };
  auto e_exp = [&](auto ic, auto& sp) {
    constexpr int i = decltype(ic)::value;
    float e = __builtin_amdgcn_exp2f(sget<i>(sp));
    asm volatile("" : "+v"(e));
    sset<i>(sp, e);
  };
  auto e_pack = [&](auto pqc, auto& sp) {
    constexpr int pq = decltype(pqc)::value;
    int pk = M::pack(sget<2 * pq>(sp), sget<2 * pq + 1>(sp));
    asm volatile("" : "+v"(pk));
    pk_prev = pk;
    pf[pq >> 2][2 * (pq & 3)] = (short)(pk & 0xffff);
    pf[pq >> 2][2 * (pq & 3) + 1] = (short)((unsigned)pk >> 16);
  };
  auto e_add = [&]() {
    psum = M::add2(pk_prev, psum);  // the row sum takes the ROUNDED weights (numerator and denominator round alike)
    asm volatile("" : "+v"(psum));
  };
  auto e_fin = [&]() {
    l_run += psum;
    asm volatile("" : "+v"(l_run));
  };
  auto exp_op = [&](auto kc, auto& sp) {
    constexpr int k = decltype(kc)::value;  // gap - 1
    constexpr int pr = k / 3, r = k % 3;
    if constexpr (pr < 8 && r == 0) e_fma(std::integral_constant<int, pr>{}, sp);
    if constexpr (pr < 8 && r == 1) e_exp(std::integral_constant<int, 2 * pr>{}, sp);
    if constexpr (pr < 8 && r == 2) e_exp(std::integral_constant<int, 2 * pr + 1>{}, sp);
    // the previous pair (pq): rounded and packed one gap after its second exponential, added to the row sum a gap later
    if constexpr (k >= 4 && k <= 25 && r == 1) e_pack(std::integral_constant<int, pr - 1>{}, sp);
    if constexpr (k >= 5 && k <= 26 && r == 2) e_add();
    if constexpr (k == 27) e_fin();
  };
  constexpr int kExpOps = 28;
  // kQ16: the same pieces over the gaps of the 16-wide QK^T phase (an MFMA of that shape leaves 8 of its 16 cycles to the
  // vector issue: one exponential, or two 4-cycle instructions, per gap). Only the gap behind the MFMA of row block 0 of a
  // fragment carries vector work (h = 0; the other one has the K read): fragment n = 4 pr + 1 + ph - ph 0 the multiply-adds
  // of pair pr, ph 1 / 2 its exponentials, ph 3 the row-sum step of pair pr - 1 and the rounding of pair pr.
  auto exp_gap16 = [&](auto nc, auto hc, auto& sp) {
    constexpr int n = decltype(nc)::value, h = decltype(hc)::value;
    if constexpr (n >= 1 && h == 0) {
      constexpr int pr = (n - 1) / 4, ph = (n - 1) % 4;
      if constexpr (pr < 8 && ph == 0) e_fma(std::integral_constant<int, pr>{}, sp);
      if constexpr (pr < 8 && ph == 1) e_exp(std::integral_constant<int, 2 * pr>{}, sp);
      if constexpr (pr < 8 && ph == 2) e_exp(std::integral_constant<int, 2 * pr + 1>{}, sp);
      if constexpr (pr >= 1 && pr < 8 && ph == 3) e_add();
      if constexpr (pr < 8 && ph == 3) e_pack(std::integral_constant<int, pr>{}, sp);
      if constexpr (n == 33) e_add();
      if constexpr (n == 34) e_fin();
    }
  };
  // first half: row maximum of the raw logits in sc, then the lazy reference: it moves only when a weight would pass
  // 2^kLazy, then to 2^kHead above the running maximum; all rows of the wave move together (one rescale pass serves them
  // all). Leaves alpha / upd for the rescale of O and takes l_run to the new reference.
  auto max_op = [&](auto kc, auto& sc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k == 0) {
      mt = fmaxf(fmaxf(sget<0>(sc), sget<1>(sc)), sget<2>(sc));
      mb = fmaxf(fmaxf(sget<3>(sc), sget<4>(sc)), sget<5>(sc));
    } else if constexpr (k == 1) {
      mt = fmaxf(fmaxf(mt, sget<6>(sc)), sget<7>(sc));
      mb = fmaxf(fmaxf(mb, sget<8>(sc)), sget<9>(sc));
    } else if constexpr (k == 2) {
      mt = fmaxf(fmaxf(mt, sget<10>(sc)), sget<11>(sc));
      mb = fmaxf(fmaxf(mb, sget<12>(sc)), sget<13>(sc));
    } else if constexpr (k == 3) {
      mt = fmaxf(fmaxf(mt, sget<14>(sc)), sget<15>(sc));
      mt = fmaxf(mt, mb);
    } else if constexpr (k == 4) {  // the other 16 tokens of this row live in lane ^ 32
      float c0 = mt, c1 = mt;
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c0), "+v"(c1));  // c0 = low half twice, c1 = high half twice
      mt = fmaxf(c0, c1);
    } else if constexpr (k == 5) {
      upd = __any((mt - m_ref) * sl2 > kLazy);  // (first tile: +inf; nothing but masked keys so far: NaN -> false)
      m_run = fmaxf(m_run, mt);
    } else if constexpr (k == 6) {
      const float cand = upd ? m_run + head_raw : -INFINITY;  // (selects, no branch in the MFMA stream)
      const float m_new = fmaxf(m_ref, cand);
      alpha = m_new > m_ref ? __builtin_amdgcn_exp2f((m_ref - m_new) * sl2 * 0.5f) : 1.0f;  // (the factor's square root)
      m_ref = m_new;
    } else if constexpr (k == 7) {
      mneg = m_ref == -INFINITY ? 0.f : -m_ref * sl2;
      l_run = l_run * alpha * alpha;
      psum = 0.f;
    }
    asm volatile("" : "+v"(mt), "+v"(mb), "+v"(alpha), "+v"(mneg), "+v"(l_run), "+v"(m_ref), "+v"(m_run));
  };
  // kQ16: the row swap in front of the maxima - two v_permlane16_swap per step (see S4): odd 16-lane rows of the rb = 0
  // register <-> even rows of the rb = 1 register
  auto swap_op = [&](auto kc, S4& sc) {
    constexpr int k = decltype(kc)::value;  // 0..3: registers r = 2 (k & 1), + 1 of token block k >> 1
    constexpr int tb = k >> 1, r0 = 2 * (k & 1);
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3"
                 : "+v"(sc.q[2 * tb][r0]), "+v"(sc.q[2 * tb][r0 + 1]), "+v"(sc.q[2 * tb + 1][r0]), "+v"(sc.q[2 * tb + 1][r0 + 1]));
  };
  constexpr int kSwapOps = kQ16 ? 4 : 0;
  constexpr int kMaxOps = 8;

  // ---- rings. K: fragment n + 3 is requested in the gap behind MFMA n; V^T: four fragments (eight transposed reads) ahead,
  // the first three requested in the last three gaps of the QK^T phase in front (kPre).
  typedef int v4i __attribute__((ext_vector_type(4)));
  static_assert(kKA >= 2 && kKA <= 6 && kVA >= 3 && kVA <= 8, "ring depths");
  v2i vlo[kVA], vhi[kVA];
  uint32_t vb[2];  // stage-relative bases of the tile whose P . V comes next (token halves e = 0, 1)
  uint32_t va0 = 0, va1 = 0;
  // order of the 32 P.V MFMAs: m -> column x = m / 8 of the dim tiles (dt % 4), then dt / 4 = (m % 8) / 2, k-step m % 2: the
  // two address registers of a column serve eight MFMAs, everything else is an immediate
  // (a capture-less lambda behind a macro: clang rejects asm outputs that name captured arrays inside nested generic lambdas)
  auto v_issue_ = [](auto mc, v2i& lo, v2i& hi, uint32_t& a0, uint32_t& a1, const uint32_t b0, const uint32_t b1) {
    constexpr int m = decltype(mc)::value;
    constexpr int q = m & 7, c = (q >> 1) * 8192 + (q & 1) * 4096;
    if constexpr ((m & 7) == 0) {  // (first use of column x: its two addresses)
      constexpr uint32_t xo = (uint32_t)((m >> 3) << 6);
      a0 = b0 ^ xo;
      a1 = b1 ^ xo;
      asm volatile("" : "+v"(a0), "+v"(a1));
    }
    if constexpr (kProbe != 1 && kProbe != 3 && kProbe != 5) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "i"(c));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "i"(c));
    }
  };
#define SGLK_V_ISSUE(M_) v_issue_(std::integral_constant<int, (M_)>{}, vlo[(M_) % kVA], vhi[(M_) % kVA], va0, va1, vb[0], vb[1])

  // ---- S^T of tile j (raw logits) in VGPRs: asm MFMAs behind hand-counted waits, asm K reads. k-step order n -> chunk pair
  // n / 4 of column block n % 4 (n < 32: one address serves four reads), then the four rope steps.
  // The first MFMA takes the constant 0 as its addend: no VALU write feeds an asm MFMA (the hazard recogniser cannot see one).
  auto qk_tile = [&](int j, v16f& s, auto with_exp, v16f& sp, const TileSrc& nxt, int jv) {
    constexpr bool kExp = decltype(with_exp)::value;
    const uint32_t sb = lds_base + (uint32_t)((j & 3) * kStageBytes);
    uint32_t kbs = kbase + sb, rbs = rbase + sb;  // (stage bases are multiples of 256: the xor of bits 5..7 commutes with the add)
    asm volatile("" : "+v"(kbs), "+v"(rbs));
    v4i kr[kKA], qr[4];  // (K ring; the four rope fragments of Q, read from LDS with their K fragments)
    uint32_t ka = kbs;
    // (capture-less, behind a macro: see v_issue_)
    auto k_issue_ = [](auto nc, v4i& kslot, v4i& qslot, uint32_t& ka_, const uint32_t kbs_, const uint32_t rbs_,
                       const uint32_t qpe_) {
      constexpr int n = decltype(nc)::value;
      if constexpr (n < 32) {
        if constexpr ((n & 3) == 0 && n > 0) {
          ka_ = kbs_ ^ (uint32_t)((n >> 2) << 5);
          asm volatile("" : "+v"(ka_));
        }
        if constexpr (kProbe != 2 && kProbe != 3 && kProbe != 5)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kslot) : "v"(ka_), "i"((n & 3) * 8192));
      } else {
        uint32_t ra = rbs_ ^ (uint32_t)((n - 32) << 5);
        if constexpr (kProbe != 2 && kProbe != 3 && kProbe != 5) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(kslot) : "v"(ra));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qslot) : "v"(qpe_), "i"((n - 32) * 1024));
        }
      }
    };
#define SGLK_K_ISSUE(N_) k_issue_(std::integral_constant<int, (N_)>{}, kr[(N_) % kKA], qr[(N_) & 3], ka, kbs, rbs, qpe_addr)
    static_for<0, kKA>([&](auto nc) { constexpr int n0 = decltype(nc)::value; SGLK_K_ISSUE(n0); });
    // under the latency of the first fragments: the rope piece of tile j + 2
    if constexpr (kExp) dma_piece(nxt, std::integral_constant<int, 8>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 36>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int ks = n < 32 ? 8 * (n & 3) + (n >> 2) : 0;  // the Q fragment that goes with K fragment n (rope: from LDS)
      // reads younger than step n's at this point: those of the steps n+1, n+2 (one each, two for a rope step: K and Q) and
      // the V^T reads of the gaps 33, 34 (two each)
      constexpr int last = n + kKA - 1 < 35 ? n + kKA - 1 : 35;  // steps n+1 .. last are in flight behind step n
      constexpr int n_lat = last - n, n_rope = last >= 32 ? last - (n + 1 > 32 ? n + 1 : 32) + 1 : 0;
      constexpr int younger = n_lat + (n_rope > 0 ? n_rope : 0) + (kExp ? (n == 34 ? 2 : n == 35 ? 4 : 0) : 0);
      static_assert(younger <= 15, "lgkmcnt is a 4-bit field");
      constexpr int wcnt = (kProbe == 2 || kProbe == 3 || kProbe == 5) ? 0 : younger;
      const v8s kf = __builtin_bit_cast(v8s, kr[n % kKA]);
      const v8s qb = n < 32 ? qf[ks] : __builtin_bit_cast(v8s, qr[n & 3]);
      if constexpr (n == 0) {
        if constexpr (std::is_same<T, bf16>::value)
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(kf), "v"(qb), "i"(wcnt));
        else
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(kf), "v"(qb), "i"(wcnt));
      } else {
        if constexpr (std::is_same<T, bf16>::value)
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "v"(qb), "i"(wcnt));
        else
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "v"(qb), "i"(wcnt));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kExp && n >= 1 && n - 1 < kExpOps && kProbe != 6) exp_op(std::integral_constant<int, n - 1>{}, sp);
      // (the slot stays reserved up to its refill: the MFMA issued just before is still reading it)
      asm volatile("" ::"v"(kr[n % kKA]));
      if constexpr (n >= 32) asm volatile("" ::"v"(qr[n & 3]));
      if constexpr (n + kKA < 36) SGLK_K_ISSUE(n + kKA);
      if constexpr (kExp) {  // P . V of tile jv comes next: its bases, then its first four fragments
        if constexpr (n == 31) {
          const uint32_t sv = lds_base + (uint32_t)((jv & 3) * kStageBytes);
          vb[0] = vbA + sv;
          vb[1] = vbB + sv;
          asm volatile("" : "+v"(vb[0]), "+v"(vb[1]));
        }
        if constexpr (n == 33) SGLK_V_ISSUE(0);
        if constexpr (n == 34) SGLK_V_ISSUE(1);
        if constexpr (n == 35) SGLK_V_ISSUE(2);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("" : "+v"(s));
#pragma unroll
    for (int i = 0; i < kKA; ++i) asm volatile("" ::"v"(kr[i]));  // (the ring stays reserved past the last MFMAs)
#undef SGLK_K_ISSUE
  };
  // ---- kQ16: the same phase on v_mfma_f32_16x16x32 (see S4). Fragment n < 32: k-step 4 cb + ks' with ks' = n / 8,
  // cb = (n / 2) % 4, token block n % 2 - one address (kbs ^ 64 ks') serves eight reads; n >= 32: rope step (n - 32) / 2,
  // token block n % 2, the two rope fragments of Q (row blocks 0, 1) requested in front of the step's first K fragment.
  auto qk_tile16 = [&](int j, S4& s, auto with_exp, S4& sp, const TileSrc& nxt, int jv) {
    constexpr bool kExp = decltype(with_exp)::value;
    const uint32_t sb = lds_base + (uint32_t)((j & 3) * kStageBytes);
    uint32_t kbs = kbase + sb, rbs = rbase + sb;
    asm volatile("" : "+v"(kbs), "+v"(rbs));
    v4i kr[kKA], qr[4];
    uint32_t ka = kbs;
    auto k_issue_ = [](auto nc, v4i& kslot, v4i& q0, v4i& q1, uint32_t& ka_, const uint32_t kbs_, const uint32_t rbs_,
                       const uint32_t qpe_) {
      constexpr int n = decltype(nc)::value;
      if constexpr (n < 32) {
        if constexpr ((n & 7) == 0 && n > 0) {
          ka_ = kbs_ ^ (uint32_t)((n >> 3) << 6);
          asm volatile("" : "+v"(ka_));
        }
        if constexpr (kProbe != 2 && kProbe != 3 && kProbe != 5)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kslot) : "v"(ka_), "i"(((n >> 1) & 3) * 8192 + (n & 1) * 4096));
      } else {
        constexpr int k2 = (n - 32) >> 1;
        uint32_t ra = rbs_ ^ (uint32_t)(k2 << 6);
        if constexpr (kProbe != 2 && kProbe != 3 && kProbe != 5) {
          if constexpr ((n & 1) == 0) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q0) : "v"(qpe_), "i"((2 * k2) * 1024));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q1) : "v"(qpe_), "i"((2 * k2 + 1) * 1024));
          }
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kslot) : "v"(ra), "i"((n & 1) * 2048));
        }
      }
    };
#define SGLK_K_ISSUE(N_) \
  k_issue_(std::integral_constant<int, (N_)>{}, kr[(N_) % kKA], qr[(N_) & 2], qr[((N_) & 2) + 1], ka, kbs, rbs, qpe_addr)
    static_for<0, kKA>([&](auto nc) { constexpr int n0 = decltype(nc)::value; SGLK_K_ISSUE(n0); });
    // under the latency of the first fragments: the rope piece of tile j + 2
    if constexpr (kExp) dma_piece(nxt, std::integral_constant<int, 8>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 36>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int tb = n & 1;
      constexpr int ks = n < 32 ? 4 * ((n >> 1) & 3) + (n >> 3) : 0;  // the latent k-step of fragment n
      constexpr int k2 = n < 32 ? 0 : (n - 32) >> 1;
      // LDS reads younger than fragment n's at this point: the fragments n+1 .. last (one read each, plus the two Q reads in
      // front of the fragments 32 and 34) and the V^T reads requested behind the fragments 33, 34 (two each)
      constexpr int last = n + kKA - 1 < 35 ? n + kKA - 1 : 35;
      constexpr int younger = (last - n) + ((n + 1 <= 32 && 32 <= last) ? 2 : 0) + ((n + 1 <= 34 && 34 <= last) ? 2 : 0) +
                              (kExp ? (n == 34 ? 2 : n == 35 ? 4 : 0) : 0);
      static_assert(younger <= 15, "lgkmcnt is a 4-bit field");
      constexpr int wcnt = (kProbe == 2 || kProbe == 3 || kProbe == 5) ? 0 : younger;
      const v8s kf = __builtin_bit_cast(v8s, kr[n % kKA]);
      const v8s qb0 = n < 32 ? qf[ks] : __builtin_bit_cast(v8s, qr[2 * k2]);
      const v8s qb1 = n < 32 ? qf[16 + ks] : __builtin_bit_cast(v8s, qr[2 * k2 + 1]);
      // (the first MFMA of a block takes the constant 0 as its addend: no VALU write feeds an asm MFMA)
      if constexpr (n < 2) {
        if constexpr (std::is_same<T, bf16>::value)
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s.q[2 * tb]) : "v"(kf), "v"(qb0), "i"(wcnt));
        else
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(s.q[2 * tb]) : "v"(kf), "v"(qb0), "i"(wcnt));
      } else {
        if constexpr (std::is_same<T, bf16>::value)
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s.q[2 * tb]) : "v"(kf), "v"(qb0), "i"(wcnt));
        else
          asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s.q[2 * tb]) : "v"(kf), "v"(qb0), "i"(wcnt));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kExp && kProbe != 6) exp_gap16(nc, std::integral_constant<int, 0>{}, sp);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (n < 2) {
        if constexpr (std::is_same<T, bf16>::value)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s.q[2 * tb + 1]) : "v"(kf), "v"(qb1));
        else
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(s.q[2 * tb + 1]) : "v"(kf), "v"(qb1));
      } else {
        if constexpr (std::is_same<T, bf16>::value)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s.q[2 * tb + 1]) : "v"(kf), "v"(qb1));
        else
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s.q[2 * tb + 1]) : "v"(kf), "v"(qb1));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kExp && kProbe != 6) exp_gap16(nc, std::integral_constant<int, 1>{}, sp);
      // (the slot stays reserved up to its refill: the MFMA issued just before is still reading it)
      asm volatile("" ::"v"(kr[n % kKA]));
      if constexpr (n >= 32) asm volatile("" ::"v"(qr[2 * k2]), "v"(qr[2 * k2 + 1]));
      if constexpr (n + kKA < 36) SGLK_K_ISSUE(n + kKA);
      if constexpr (kExp) {  // P . V of tile jv comes next: its bases, then its first three fragments
        if constexpr (n == 31) {
          const uint32_t sv = lds_base + (uint32_t)((jv & 3) * kStageBytes);
          vb[0] = vbA + sv;
          vb[1] = vbB + sv;
          asm volatile("" : "+v"(vb[0]), "+v"(vb[1]));
        }
        if constexpr (n == 33) SGLK_V_ISSUE(0);
        if constexpr (n == 34) SGLK_V_ISSUE(1);
        if constexpr (n == 35) SGLK_V_ISSUE(2);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    s_pin(s);
#pragma unroll
    for (int i = 0; i < kKA; ++i) asm volatile("" ::"v"(kr[i]));  // (the ring stays reserved past the last MFMAs)
#undef SGLK_K_ISSUE
  };
  auto mask_tile = [&](int j, auto& s) {  // (rare, uniform: keys past a row's horizon)
    const int t = t_begin + j;
    if (t * kTile + kTile > kv_first || t * kTile + kTile > seq) {
      s_settle(s);
      if constexpr (kQ16) {  // (in front of the row swap: block (tb, rb), register r = token 16 tb + 4 g(kg) + r of row 16 rb + j)
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int ln = tl & 63, gk = (0x78 >> (2 * (ln >> 4))) & 3;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int r_tok = (wave * 32 + 16 * rb + (ln & 15)) >> p.hp_shift;
          const int kv_row = (cu_seqlens_q != nullptr && p.causal && r_tok < n_tok) ? kv_first + r_tok : seq;
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (t * kTile + 16 * tb + 4 * gk + r >= kv_row) s.q[2 * tb + rb][r] = -INFINITY;
        }
      } else {
        const RowId r = row_id();  // (recomputed here: nothing lane-dependent of the epilogue stays live across the tile loop)
        const int kv_row = (cu_seqlens_q != nullptr && p.causal && r.my_tok < n_tok) ? kv_first + r.my_tok : seq;  // the row's horizon
#pragma unroll
        for (int v = 0; v < 16; ++v)
          if (t * kTile + (v & 3) + 8 * (v >> 2) + 4 * r.u >= kv_row) s[v] = -INFINITY;
      }
    }
  };

  // ---- O^T += V^T . P^T of tile jv: asm MFMAs on the fixed accumulators and asm transposed reads in program order with
  // hand-counted waits. kPre: the first four fragments were requested by the QK^T phase in front. kSlot >= 0: this wave's
  // DMA pieces 0..7 of the tile after next go out in the gaps m % 4 == kSlot (wave w: slot w - one wave at a time on the
  // CU's vector-memory path).
  auto pv_tile = [&](int jv, auto pre, auto with_max, auto& sc, auto slot, const TileSrc& nxt) {
    constexpr bool kPre = decltype(pre)::value, kMax = decltype(with_max)::value;
    constexpr int kSlot = decltype(slot)::value;
    if constexpr (!kPre) {
      const uint32_t sv = lds_base + (uint32_t)((jv & 3) * kStageBytes);
      vb[0] = vbA + sv;
      vb[1] = vbB + sv;
      asm volatile("" : "+v"(vb[0]), "+v"(vb[1]));
      static_for<0, kVA>([&](auto mc) { constexpr int m0 = decltype(mc)::value; SGLK_V_ISSUE(m0); });
    } else {  // (three fragments came with the QK^T phase: its K ring was still live there)
      static_for<3, kVA>([&](auto mc) { constexpr int m0 = decltype(mc)::value; SGLK_V_ISSUE(m0); });
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 32>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int q = m & 7, dt = (m >> 3) + 4 * (q >> 1), ss = q & 1;
      constexpr int ahead = (32 - m < kVA ? 32 - m : kVA) - 1;  // fragments issued after this one
      // (the two halves are coalesced into the 4-register operand in place - no copy may sit between the reads and the
      //  wait the MFMA statement starts with; the ISA shows none)
      const v8s f = __builtin_bit_cast(v8s, __builtin_shufflevector(vlo[m % kVA], vhi[m % kVA], 0, 1, 2, 3));
      constexpr int wcnt = (kProbe == 1 || kProbe == 3 || kProbe == 5) ? 0 : 2 * ahead;
      if constexpr (kProbe == 8) {  // (timing probe, garbage results: the phase's FLOP on the 16-wide shape - two MFMAs per fragment)
        asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(f), "v"(pf[ss]),
                     "i"(dt * 16), "i"(dt * 16 + 3), "i"(wcnt));
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(f), "v"(pf[ss]), "i"(dt * 16 + 8), "i"(dt * 16 + 11));
      } else {
        M::template wait_acc_agpr<dt * 16, wcnt>(f, pf[ss]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kMax && kProbe != 6) {  // (kQ16: the row swap of the tile's raw logits first)
        if constexpr (kQ16 && m >= 2 && m - 2 < kSwapOps) swap_op(std::integral_constant<int, m - 2>{}, sc);
        if constexpr (m >= 2 + kSwapOps && m - 2 - kSwapOps < kMaxOps) max_op(std::integral_constant<int, m - 2 - kSwapOps>{}, sc);
      }
      // the fragment's registers stay reserved past the micro-op (a VALU write into an operand of the MFMA issued just
      // before is not interlocked); the refill of this slot lands tens of cycles later
      asm volatile("" ::"v"(f));
      if constexpr (m + kVA < 32) SGLK_V_ISSUE(m + kVA);
      if constexpr (kSlot >= 0 && (m & 3) == kSlot) dma_piece(nxt, std::integral_constant<int, (m >> 2)>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // the operands of the last MFMAs stay reserved until the matrix pipe has read them
    asm volatile("" ::"v"(pf[0]), "v"(pf[1]));
#pragma unroll
    for (int i = 0; i < kVA; ++i) asm volatile("" ::"v"(vlo[i]), "v"(vhi[i]));
    __builtin_amdgcn_sched_barrier(0);
  };
#undef SGLK_V_ISSUE

  unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
  auto stamp = [&](int k) {
    if constexpr (kStamp) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      st_sum[k] += now - st_t;
      st_t = now;
    }
  };
  unsigned long long clk0 = 0, rt0 = 0;
  if constexpr (kStamp) {
    st_t = __builtin_amdgcn_s_memtime();
    clk0 = st_t;
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  // One iteration: tile j's QK^T (+ exponentials of tile j-1 from s_prv, + the first V^T reads of tile j-1), then P.V of
  // tile j-1 (+ maxima of tile j in s_cur, + the DMA pieces of tile j+2).
  auto iter = [&](int j, auto& s_cur, auto& s_prv) {
    stamp(8);
    asm volatile("s_waitcnt vmcnt(9)" : "+v"(v_ids)::"memory");  // tile j landed; so has a window fetched an iteration ago
    stamp(0);
    __builtin_amdgcn_s_barrier();  // tile j landed for every wave; every wave is done with tile j-2
    stamp(1);
    int pga, pgb;
    ids_of(tile_of(j + 2), pga, pgb);
    ids_advance(tile_of(j + 3));  // (in front of this iteration's nine pieces)
    const TileSrc nxt = tile_src(tile_of(j + 2), (j + 2) & 3, pga, pgb);
    stamp(2);
    if (work) {
      // + exponentials of tile j-1 -> pf
      if constexpr (kQ16) qk_tile16(j, s_cur, std::true_type{}, s_prv, nxt, j - 1);
      else qk_tile(j, s_cur, std::true_type{}, s_prv, nxt, j - 1);
      mask_tile(j, s_cur);
      __builtin_amdgcn_sched_barrier(0);
      // (no stamp between the phases: s_memtime is a scalar memory read, its wait would drain the V^T reads in flight here)
      // + row maxima / reference of tile j; the slot of this wave's DMA pieces is a template parameter: four copies
      if (wave == 0) pv_tile(j - 1, std::true_type{}, std::true_type{}, s_cur, std::integral_constant<int, 0>{}, nxt);
      else if (wave == 1) pv_tile(j - 1, std::true_type{}, std::true_type{}, s_cur, std::integral_constant<int, 1>{}, nxt);
      else if (wave == 2) pv_tile(j - 1, std::true_type{}, std::true_type{}, s_cur, std::integral_constant<int, 2>{}, nxt);
      else pv_tile(j - 1, std::true_type{}, std::true_type{}, s_cur, std::integral_constant<int, 3>{}, nxt);
      stamp(4);
      if (upd) {  // rare: the reference moved, rescale O (AGPR -> VGPR -> AGPR) before the next P . V
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");  // (MFMA -> v_accvgpr_read wait states)
        static_for<0, 32>([&](auto ic) { agpr_scale8_sq<decltype(ic)::value * 8>(alpha); });
        asm volatile("s_nop 7");
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      static_for<0, 9>([&](auto ic) { dma_piece(nxt, ic); });
    }
  };
  typename std::conditional<kQ16, S4, v16f>::type s_a, s_b;
  {  // tile 0: nothing to overlap with
    stamp(5);
    asm volatile("s_waitcnt vmcnt(9)" : "+v"(v_ids)::"memory");
    __builtin_amdgcn_s_barrier();
    int pga, pgb;
    ids_of(tile_of(2), pga, pgb);
    ids_advance(tile_of(3));
    const TileSrc nxt = tile_src(tile_of(2), 2, pga, pgb);
    static_for<0, 9>([&](auto ic) { dma_piece(nxt, ic); });
    if (work) {
      if constexpr (kQ16) qk_tile16(0, s_b, std::false_type{}, s_b, nxt, 0);
      else qk_tile(0, s_b, std::false_type{}, s_b, nxt, 0);
      mask_tile(0, s_b);
      s_settle(s_b);  // (asm MFMA result -> VALU read wait states)
      if constexpr (kQ16) {
        static_for<0, kSwapOps>([&](auto kc) { swap_op(kc, s_b); });
        asm volatile("s_nop 1");
      }
      static_for<0, kMaxOps>([&](auto kc) { max_op(kc, s_b); });  // (O is zero: nothing to rescale)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  int j = 1;
  for (; j + 1 < n_my; j += 2) {  // (two tiles per trip: the logits of the tile in flight and of the one before swap roles)
    iter(j, s_a, s_b);
    iter(j + 1, s_b, s_a);
  }
  const bool odd_tail = j < n_my;
  if (odd_tail) iter(j, s_a, s_b);
  if (work) {
    if (odd_tail) static_for<0, kExpOps>([&](auto kc) { exp_op(kc, s_a); });
    else static_for<0, kExpOps>([&](auto kc) { exp_op(kc, s_b); });
    __builtin_amdgcn_sched_barrier(0);
    v16f dummy;
    TileSrc none = {};
    pv_tile(n_my - 1, std::false_type{}, std::false_type{}, dummy, std::integral_constant<int, -1>{}, none);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");  // (MFMA -> v_accvgpr_read wait states of the epilogue)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the duplicate tiles: no LDS-DMA may be in flight when the workgroup ends)
  stamp(4);
#ifdef SGLK_PROBES
  auto write_stamps = [&]() {
    if constexpr (kStamp) {
      stamp(6);
      if (lane == 0) {
        unsigned long long* d = g_mla_stamps + (((size_t)b * p.splits + split) * 4 + wave) % 4096 * 16;
        for (int k = 0; k < 12; ++k) d[k] = st_sum[k];
        d[12] = __builtin_amdgcn_s_memtime() - clk0;       // shader cycles of the whole tile loop + epilogue
        d[13] = __builtin_amdgcn_s_memrealtime() - rt0;    // the same in 100 MHz ticks
        d[15] = (unsigned long long)n_my;
      }
    }
  };
#else
  auto write_stamps = [&]() {};
#endif

  // ---- epilogue: O^T tile dt, register v: dim 32 dt + 8 (v / 4) + 4 u + v % 4 of this lane's row (see mla_stage_group)
  if (p.probe == 1) return;
  const RowId rid = row_id();
  const int l31b = lane & 31;
  float c0 = l_run, c1 = l_run;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c0), "+v"(c1));
  const float l_tot = c0 + c1;
  const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (p.splits == 1) {
    __syncthreads();  // (every wave has left the tile loop and drained its DMA: the stages are free)
    static_for<0, 64>([&](auto ic) {
      constexpr int i = decltype(ic)::value;  // dim tile i / 4, register group i % 4
      const v4f v = agpr_read4<i * 4>();
      const v4f o = {v[0] * inv_l, v[1] * inv_l, v[2] * inv_l, v[3] * inv_l};
      mla_stage_group<M>(smem, wave, l31b, rid.u, i, o);
    });
    mla_flush_rows<T>(p, smem, wave, lane, q_row0, n_tok);
  } else {
    const float my_lse = l_tot > 0.f ? m_ref * sl2 + log2f(l_tot) : -INFINITY;
    if (!mla_take_ticket(p, b, smem, tid)) {
      v4f* wo = reinterpret_cast<v4f*>(p.ws_o) + mla_slab_f4(p, b, split, wave) + lane;  // fragment order: see above
      static_for<0, 64>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const v4f v = agpr_read4<i * 4>();
        const v4f w = {v[0] * inv_l, v[1] * inv_l, v[2] * inv_l, v[3] * inv_l};
        v4f* const dst = wo + i * 64;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(w) : "memory");
      });
      if (rid.ok && rid.u == 0)
        __hip_atomic_store(p.ws_lse + ((int64_t)b * p.splits + split) * H + rid.my_head, my_lse, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      mla_count_in(p, b, tid);
    } else {
      const bool all_in = mla_wait_others(p, b, smem, tid);
      mla_merge_splits<T, M, true>(p, b, split, H, rid.ok ? rid.my_head : 0, smem, wave, lane, my_lse, inv_l, all_in);
      mla_flush_rows<T>(p, smem, wave, lane, q_row0, n_tok);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  write_stamps();
}

// out[b,h,:] = sum_s w_s O_s / sum_s w_s,  w_s = 2^(lse_s - max lse)
template <typename T>
__global__ __launch_bounds__(128) void mla_reduce_kernel(T* __restrict__ out, const float* __restrict__ ws_o,
                                                         const float* __restrict__ ws_lse, int H, int splits) {
  const int h = blockIdx.x, b = blockIdx.y;
  const float* lse = ws_lse + (int64_t)b * splits * H + h;
  float mx = -INFINITY;
  for (int s = 0; s < splits; ++s) mx = fmaxf(mx, lse[(int64_t)s * H]);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  float wsum = 0.f;
  const int d = threadIdx.x * 4;
  for (int s = 0; s < splits; ++s) {
    const float l = lse[(int64_t)s * H];
    const float w = (l == -INFINITY) ? 0.f : exp2f(l - mx);
    if (w != 0.f) {
      const v4f v = *reinterpret_cast<const v4f*>(ws_o + (((int64_t)b * splits + s) * H + h) * kLatent + d);
      acc[0] += w * v[0]; acc[1] += w * v[1]; acc[2] += w * v[2]; acc[3] += w * v[3];
      wsum += w;
    }
  }
  const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
  Vec<T, 4> o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (T)(acc[i] * inv);
  store_vec<T, 4>(out + ((int64_t)b * H + h) * kLatent + d, o);
}

template <typename T, int W>
static int launch_w(hipStream_t st, const MlaParams& p, int B, const void* q_nope, const void* q_pe, const void* cache,
                    const int32_t* seq_lens, const int32_t* page_table, const int32_t* cu_seqlens_q = nullptr,
                    int token_blocks = 1) {
  static unsigned long long attr_done = 0;
  constexpr int lds = Cfg<W>::kLdsBytes;
  if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&mla_decode_kernel<T, W>), lds, &attr_done, "flash_mla_decode"))
    return rc;
  mla_decode_kernel<T, W><<<dim3(p.splits, B, token_blocks), kThreads, lds, st>>>(
      p, (const T*)q_nope, (const T*)q_pe, (const char*)cache, seq_lens, page_table, cu_seqlens_q);
  return check_launch(cu_seqlens_q ? "flash_mla_prefill" : "flash_mla_decode");
}

#ifdef SGLK_PROBES
static int g_mla_variant = 0;  // kbench: 70 + probe = the tile loop with in-kernel stamps, 200 + probe = without (wall time)
#endif
template <typename T>
static int launch_rows128x(hipStream_t st, const MlaParams& p, int B, const void* q_nope, const void* q_pe,
                           const void* cache, const int32_t* seq_lens, const int32_t* page_table,
                           const int32_t* cu_seqlens_q = nullptr, int token_blocks = 1) {
  const dim3 grid(p.splits, B, token_blocks);
  constexpr int ldsz = 4 * kStageBytes + 16384;  // (the four stages + the rope part of Q = all 160 KiB)
#define SGLK_MLA_LAUNCH(KERNEL, lds)                                                                                      \
  {                                                                                                                    \
    static unsigned long long attr = 0;                                                                                \
    if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&KERNEL), lds, &attr, "flash_mla_decode")) return rc;   \
    KERNEL<<<grid, kThreads2, lds, st>>>(p, (const T*)q_nope, (const T*)q_pe, (const char*)cache, seq_lens, page_table, \
                                         cu_seqlens_q);                                                                \
    return check_launch(cu_seqlens_q ? "flash_mla_prefill" : "flash_mla_decode");                                      \
  }
#ifdef SGLK_PROBES
  if constexpr (std::is_same<T, bf16>::value) {
    if (g_mla_variant == 70) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, true, 0>), ldsz)
    if (g_mla_variant == 73) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, true, 3>), ldsz)
    if (g_mla_variant == 74) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, true, 4>), ldsz)
    if (g_mla_variant == 75) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, true, 5>), ldsz)
    if (g_mla_variant == 76) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, true, 6>), ldsz)
    if (g_mla_variant == 203) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 3>), ldsz)  // probes without stamps: wall time
    if (g_mla_variant == 204) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 4>), ldsz)
    if (g_mla_variant == 205) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 5>), ldsz)
    if (g_mla_variant == 206) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 6>), ldsz)
    if (g_mla_variant == 207) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 7>), ldsz)
    if (g_mla_variant == 208) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 8, 3, 4, true>), ldsz)   // + P.V FLOP on 16x16x32 (garbage)
    if (g_mla_variant == 216) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 0, 3, 4, true>), ldsz)   // QK^T on 16x16x32
    if (g_mla_variant == 232) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 0, 3, 4, false>), ldsz)  // QK^T on 32x32x16
    if (g_mla_variant == 86) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, true, 0, 3, 4, true>), ldsz)     // stamped, 16-wide
    if (g_mla_variant == 87) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, true, 0, 3, 4, false>), ldsz)    // stamped, 32-wide
  }
#endif
  if (p.q16) SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 0, 3, 4, true>), ldsz)
  SGLK_MLA_LAUNCH((mla_rows128z_kernel<T, false, 0, 3, 4, false>), ldsz)
#undef SGLK_MLA_LAUNCH
}

// Diagnostic build only (-DSGLK_PROBES, tools/kbench): force the number of waves per 16-head group (0 = automatic;
// 9 = never the rows128 kernel) and the streaming-only timing probe (garbage results). The release library has
// neither switch.
#ifdef SGLK_PROBES
static int g_mla_waves_per_group = 0;
static int g_mla_probe = 0;
#else
constexpr int g_mla_waves_per_group = 0;
constexpr int g_mla_probe = 0;
#endif

template <typename T>
static int launch(hipStream_t st, const MlaParams& p, int B, const void* q_nope, const void* q_pe, const void* cache,
                  const int32_t* seq_lens, const int32_t* page_table) {
  const int ngroups = (p.H + 15) >> 4;
  int w = ngroups <= 1 ? 8 : ngroups <= 2 ? 4 : ngroups <= 4 ? 2 : 1;
  if (g_mla_waves_per_group > 0 && g_mla_waves_per_group <= w) w = g_mla_waves_per_group;
  int rc;
  // More than 64 heads: the rows128z kernel - while a batch element has at most four splits. Its merge is ONE workgroup per
  // batch element adding the other splits' 256-KiB partial results one after the other: fine for 2 .. 4 splits (batch >= 64),
  // ruinous for the 16 .. 64 splits a small batch gets - a batch-size sweep (round 5, late) found bs 1 / 4 / 16 x 8192 keys x 128 heads
  // at 520 / 587 / 186 us against 61 / 69 / 85 us for the 8-wave kernel below with its parallel reduce launch (bs 32: 149 / 128; bs 64:
  // 178 / 207; bs 128: 301 / 376).
  const bool rows128 = ngroups > 4 && g_mla_waves_per_group == 0 && p.splits <= 4;
  if (rows128) {
    rc = launch_rows128x<T>(st, p, B, q_nope, q_pe, cache, seq_lens, page_table);
  } else
  switch (w) {
    case 8: rc = launch_w<T, 8>(st, p, B, q_nope, q_pe, cache, seq_lens, page_table); break;
    case 4: rc = launch_w<T, 4>(st, p, B, q_nope, q_pe, cache, seq_lens, page_table); break;
    case 2: rc = launch_w<T, 2>(st, p, B, q_nope, q_pe, cache, seq_lens, page_table); break;
    default: rc = launch_w<T, 1>(st, p, B, q_nope, q_pe, cache, seq_lens, page_table); break;
  }
  if (rc) return rc;
  const bool merged_in_kernel = rows128;  // (the rows128z kernel merges its splits itself)
  if (p.splits > 1 && !merged_in_kernel) {
    mla_reduce_kernel<T><<<dim3(p.H, B), 128, 0, st>>>((T*)p.out, p.ws_o, p.ws_lse, p.H, p.splits);
    return check_launch("flash_mla_decode(reduce)");
  }
  return SGLK_OK;
}

}  // namespace
// bytes of the merge counters in front of the workspace: two 8-byte words per batch element, padded to 256 (keeps what
// follows aligned)
static inline int64_t mla_counter_bytes(int64_t batch) { return (batch * 16 + 255) / 256 * 256; }
// rows of partial O per (batch element, split): the rows128 kernel (H > 64) stores whole 128-row fragment slabs
static inline int64_t mla_ws_rows(int64_t H) { return H > 64 ? 128 : H; }
}  // namespace sglk

#ifdef SGLK_PROBES
extern "C" SGLK_API void sglk_debug_set_mla_waves_per_group(int w) { sglk::g_mla_waves_per_group = w; }
extern "C" SGLK_API void sglk_debug_set_mla_probe(int v) { sglk::g_mla_probe = v; }
extern "C" SGLK_API void sglk_debug_set_mla_variant(int v) { sglk::g_mla_variant = v; }
extern "C" SGLK_API int sglk_debug_get_mla_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sglk::g_mla_stamps), (size_t)n * 8);
}
#endif

// Number of KV splits used when the caller passes num_kv_splits < 1: about one workgroup per CU, and at
// least 4 tiles (128 tokens) of work per split. (The reference's set_split_kv, mla_decode.cpp:60-93, is tuned
// for its one-work-group-per-head grid on Xe2 and does not transfer.)
extern "C" int64_t sglk_mla_decode_auto_splits(int64_t batch, int64_t max_seq_len) {
  const int64_t tiles = (max_seq_len + 31) / 32;
  int64_t s = 256 / (batch > 0 ? batch : 1);
  const int64_t cap = (tiles + 3) / 4;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  // (lease zh, explicit counts against this rule: bs 1 / 4 x 8192 keys at 16 heads 64 splits - the old choice - 48.7 / 54.3 us, 32 splits
  //  37.5 / 41.4, 16: 39.0 / 43.0; at 128 heads 66.3 / 78.4 against 52.7 / 57.3: at most 32 splits unless a split would then be longer than
  //  32 tiles)
  const int64_t most = tiles / 32 > 32 ? tiles / 32 : 32;
  if (s > most) s = most;
  if (s > 128) s = 128;
  return s;
}

extern "C" int64_t sglk_mla_decode_workspace_size(int64_t max_seq_len, int64_t batch, int64_t num_heads,
                                                  int64_t num_kv_splits) {
  if (num_kv_splits < 1) num_kv_splits = sglk_mla_decode_auto_splits(batch, max_seq_len);
  if (num_kv_splits == 1) return 0;
  // {ticket, published} counter words of the in-kernel merge (a block of their own at the start), then
  // fp32 partial O and the log2-sum-exps
  return sglk::mla_counter_bytes(batch) + batch * num_kv_splits * (sglk::mla_ws_rows(num_heads) * sglk::kLatent + num_heads) * 4;
}

extern "C" int sglk_flash_mla_decode(sglk_stream_t stream, void* out, const void* q_nope, const void* q_pe,
                                     const void* cache, const int32_t* seq_lens, const int32_t* page_table,
                                     void* workspace, int64_t workspace_bytes, int64_t batch, int64_t num_heads,
                                     int64_t page_size, int64_t pages_per_seq, int64_t q_nope_stride_b,
                                     int64_t q_nope_stride_h, int64_t q_pe_stride_b, int64_t q_pe_stride_h,
                                     int64_t cache_page_stride, int64_t table_stride, float sm_scale,
                                     int64_t num_kv_splits, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(batch >= 0 && num_heads > 0 && num_heads <= 128, "flash_mla_decode: H must be in [1, 128], got %lld",
               (long long)num_heads);
  SGLK_REQUIRE(page_size == 16 || page_size == 32 || page_size == 64 || page_size == 128,
               "flash_mla_decode: Unsupported page size: %lld", (long long)page_size);
  SGLK_REQUIRE(pages_per_seq > 0, "flash_mla_decode: block num must be greater than 0");
  SGLK_REQUIRE(dtype == SGLK_BF16 || dtype == SGLK_F16, "flash_mla_decode: dtype must be Half or BFloat16");
  SGLK_REQUIRE(q_nope_stride_b % 8 == 0 && q_nope_stride_h % 8 == 0 && q_pe_stride_b % 8 == 0 &&
                   q_pe_stride_h % 8 == 0 && (uintptr_t)q_nope % 16 == 0 && (uintptr_t)q_pe % 16 == 0 &&
                   (uintptr_t)cache % 16 == 0 && cache_page_stride % 8 == 0,
               "flash_mla_decode: q and cache rows must be 16-byte aligned");
  SGLK_REQUIRE(cache_page_stride > 0 && cache_page_stride * 2 < (int64_t)1 << 32,
               "flash_mla_decode: a page stride must be below 4 GiB, got %lld elements", (long long)cache_page_stride);
  if (batch == 0) return SGLK_OK;
  const int64_t max_seq = pages_per_seq * page_size;
  int64_t splits = num_kv_splits < 1 ? sglk_mla_decode_auto_splits(batch, max_seq) : num_kv_splits;
  const int64_t max_tiles = (max_seq + 31) / 32;
  if (splits > max_tiles) splits = max_tiles;
  SGLK_REQUIRE(splits <= 32768, "flash_mla_decode: at most 32768 KV splits, got %lld", (long long)splits);
  if (splits > 1) {
    const int64_t need = mla_counter_bytes(batch) + batch * splits * (mla_ws_rows(num_heads) * kLatent + num_heads) * 4;
    SGLK_REQUIRE(workspace != nullptr && workspace_bytes >= need,
                 "flash_mla_decode: workspace too small: %lld bytes given, %lld needed for %lld splits",
                 (long long)workspace_bytes, (long long)need, (long long)splits);
    SGLK_REQUIRE((uintptr_t)workspace % 16 == 0, "flash_mla_decode: workspace must be 16-byte aligned");
  }
  MlaParams p;
  p.out = out;
  p.ws_cnt = (unsigned long long*)workspace;  // (no zeroing: the words initialise themselves, see mla_cnt_up)
  p.ws_o = workspace ? (float*)((char*)workspace + mla_counter_bytes(batch)) : nullptr;
  p.ws_lse = p.ws_o ? p.ws_o + batch * splits * mla_ws_rows(num_heads) * kLatent : nullptr;
  p.qn_sb = q_nope_stride_b;
  p.qn_sh = q_nope_stride_h;
  p.qp_sb = q_pe_stride_b;
  p.qp_sh = q_pe_stride_h;
  p.page_stride_bytes = cache_page_stride * 2;
  p.table_stride = table_stride;
  p.H = (int)num_heads;
  p.page_shift = page_size == 16 ? 4 : page_size == 32 ? 5 : page_size == 64 ? 6 : 7;
  p.splits = (int)splits;
  p.hp_shift = 7;
  p.causal = 0;
  p.probe = g_mla_probe;
  // QK^T on the 16-wide MFMA shape where the launch keeps every CU busy for long (the part then holds a higher clock under
  // it: bs 128 x 8192 -2.7 %, bs 256 x 2048 -5 %, H = 96 -3 %); short launches are issue-bound, not power-bound, and lose
  // ~1 % to its narrower gaps (bs 32 x 8192, bs 64 x 4096, bs 128 x 1024): from 48 tiles per CU on
  p.q16 = batch * max_tiles >= 48 * (int64_t)num_cus() ? 1 : 0;
  p.scale_log2 = sm_scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SGLK_BF16) return launch<bf16>(st, p, (int)batch, q_nope, q_pe, cache, seq_lens, page_table);
  return launch<f16>(st, p, (int)batch, q_nope, q_pe, cache, seq_lens, page_table);
}

// flash_mla_prefill: varlen causal MLA over the same paged latent cache (reference python/sgl_kernel/attention.py:
// 149-233, tests/test_flash_mla_prefill.py:30-90). Same kernel as the decode: a workgroup takes 128 / Hp consecutive
// query tokens of one sequence (Hp = H rounded up to 16, 32, 64 or 128) as its 128 MFMA rows and streams the cache
// once for all of them; each row masks the keys past its own causal horizon.
extern "C" int64_t sglk_flash_mla_prefill_workspace_size(int64_t, int64_t, int64_t, int64_t, int64_t) { return 0; }

extern "C" int sglk_flash_mla_prefill(sglk_stream_t stream, void* out, const void* q_nope, const void* q_pe,
                                      const void* cache, const int32_t* cu_seqlens_q, const int32_t* seq_lens_k,
                                      const int32_t* page_table, int64_t batch, int64_t max_seqlen_q,
                                      int64_t num_heads, int64_t page_size, int64_t pages_per_seq,
                                      int64_t q_nope_stride_t, int64_t q_nope_stride_h, int64_t q_pe_stride_t,
                                      int64_t q_pe_stride_h, int64_t cache_page_stride, int64_t table_stride,
                                      float sm_scale, int causal, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(batch >= 0 && num_heads > 0 && num_heads <= 128, "flash_mla_prefill: H must be in [1, 128], got %lld",
               (long long)num_heads);
  SGLK_REQUIRE(page_size == 16 || page_size == 32 || page_size == 64 || page_size == 128,
               "flash_mla_prefill: Unsupported page size: %lld", (long long)page_size);
  SGLK_REQUIRE(pages_per_seq > 0, "flash_mla_prefill: block num must be greater than 0");
  SGLK_REQUIRE(dtype == SGLK_BF16 || dtype == SGLK_F16, "flash_mla_prefill: dtype must be Half or BFloat16");
  SGLK_REQUIRE(q_nope_stride_t % 8 == 0 && q_nope_stride_h % 8 == 0 && q_pe_stride_t % 8 == 0 &&
                   q_pe_stride_h % 8 == 0 && (uintptr_t)q_nope % 16 == 0 && (uintptr_t)q_pe % 16 == 0 &&
                   (uintptr_t)cache % 16 == 0 && cache_page_stride % 8 == 0,
               "flash_mla_prefill: q and cache rows must be 16-byte aligned");
  SGLK_REQUIRE(cache_page_stride > 0 && cache_page_stride * 2 < (int64_t)1 << 32,
               "flash_mla_prefill: a page stride must be below 4 GiB, got %lld elements", (long long)cache_page_stride);
  if (batch == 0 || max_seqlen_q <= 0) return SGLK_OK;
  MlaParams p;
  p.out = out;
  p.ws_o = nullptr;
  p.ws_lse = nullptr;
  p.qn_sb = q_nope_stride_t;
  p.qn_sh = q_nope_stride_h;
  p.qp_sb = q_pe_stride_t;
  p.qp_sh = q_pe_stride_h;
  p.page_stride_bytes = cache_page_stride * 2;
  p.table_stride = table_stride;
  p.H = (int)num_heads;
  p.page_shift = page_size == 16 ? 4 : page_size == 32 ? 5 : page_size == 64 ? 6 : 7;
  p.splits = 1;
  p.hp_shift = num_heads <= 16 ? 4 : num_heads <= 32 ? 5 : num_heads <= 64 ? 6 : 7;
  p.causal = causal ? 1 : 0;
  p.probe = 0;
  p.scale_log2 = sm_scale * 1.4426950408889634f;
  const int tpw = 1 << (7 - p.hp_shift);
  const int token_blocks = (int)((max_seqlen_q + tpw - 1) / tpw);
  // (QK^T on the 16-wide MFMA shape once the grid fills the chip: 16 x 512 over 4096 keys 7.19 - 7.30 ms against 7.38 - 7.66,
  //  8 x 128 over 8192 1.815 against 1.873, 4 x 64 over 2048 0.133 against 0.136; on 96 - 128 workgroups it loses 1 - 4 %)
  p.q16 = batch * token_blocks >= (int64_t)num_cus() ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  // 128 rows per workgroup either way: the rows128 kernel (the hook value 1 selects the 8-wave kernel instead)
  if (g_mla_waves_per_group == 1) {
    if (dtype == SGLK_BF16)
      return launch_w<bf16, 1>(st, p, (int)batch, q_nope, q_pe, cache, seq_lens_k, page_table, cu_seqlens_q, token_blocks);
    return launch_w<f16, 1>(st, p, (int)batch, q_nope, q_pe, cache, seq_lens_k, page_table, cu_seqlens_q, token_blocks);
  }
  if (dtype == SGLK_BF16)
    return launch_rows128x<bf16>(st, p, (int)batch, q_nope, q_pe, cache, seq_lens_k, page_table, cu_seqlens_q, token_blocks);
  return launch_rows128x<f16>(st, p, (int)batch, q_nope, q_pe, cache, seq_lens_k, page_table, cu_seqlens_q, token_blocks);
}
