// moe_grouped_mm_nt_xe20: grouped GEMM over ragged expert row blocks with 16-bit weights.
//
// Replaces reference src/sycl/GroupGemmXe20.cpp:160-275 (host) and its CuTe kernels. Contract kept (schema
// src/torch_extension_sycl.cc:208-212; caller python/sgl_kernel/moe.py:763-775, :812-860):
//   for expert e with rows[e] consecutive rows of `activations` [total_m, K]:
//     out_e = A_e @ W_e^T (+ bias_e fp32),   W [E, N, K] (row stride ld_b), out [total_m, N]
// fuse_act (reference kernels/moe/xe20/bf16/moe_mainloop.hpp:232-247, :375-390; common/activation.hpp:31-50): the
// activation is applied to the fp32 accumulators (+ bias) in the epilogue, one rounding to T:
//   silu / gelu (gated): W holds gate rows [0, N/2) then up rows [N/2, N); out[m, n] = T(act(gate) * up), out is
//                        [total_m, N/2]. A wave's two n tiles are gate tile n and up tile n + N/2: the product never
//                        leaves the registers (the unfused route writes and re-reads a [total_m, N] intermediate).
//   relu2 (not gated):   out[m, n] = T(max(x, 0)^2), out is [total_m, N].
//
// Kernel: the skeleton of moe_w4a16.hip without the dequantisation. Block = 4 waves; wave w owns NW 16-wide n tiles
// and all MT 16-row m tiles of the block's expert rows. Weights stream HBM -> registers (lane (n, g) reads the 16
// bytes k = 32 j + 8 g .. + 7 of its row per 32-deep MFMA step: 64 contiguous bytes per row and instruction, 256 per
// 128-deep block), two blocks ahead in a static register ring; the activation tile [BM x 128] goes through LDS once
// per block for all 4 waves (loaded one block ahead, written to LDS the iteration after; LDS-only barrier).
#include <algorithm>

#include "common.h"
#include "moe_tiles.h"

namespace sglk {
// moe_persist.hip: the dense tile pipeline for prefill row counts (returns 1 after launching, 0 if the shape does not qualify)
int moe_persist_try(hipStream_t st, void* out, const void* act, const void* w, const void* scales, const void* zeros,
                    int group_shift, const float* bias,
                    const int32_t* rows, int64_t total_m, int E, int N, int K, int64_t ldb, int64_t stride_e, int dtype, int w4,
                    int fuse, float act_limit, float act_alpha = 0.f);
namespace {

static thread_local int t_tail_flag = 0;  // kMoeTailFlag while the launch covers only the rows moe_persist.hip left over
static thread_local float t_act_alpha = 0.f, t_act_limit = 0.f;  // the gpt-oss swiglu's parameters of the running call

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ v4f mma16(const v4i& a, const v4i& b, const v4f& c);
template <>
__device__ __forceinline__ v4f mma16<bf16>(const v4i& a, const v4i& b, const v4f& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ v4f mma16<f16>(const v4i& a, const v4i& b, const v4f& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
}

// FUSE_SWIGLU (5, the numbering of the W4A16 op, where 4 is the clamped swiglu): the gpt-oss swiglu of the reference's fused
// epilogue (kernels/moe/xe20/common/activation.hpp:31-42; moe_kernel.hpp:109-125: gate = weight rows 0, 2, 4, .., up = rows 1, 3,
// 5, .. - INTERLEAVED, bias likewise): out[m, n] = g * sigmoid(alpha g) * (u + 1), g = min(x[2 n], limit), u = clamp(x[2 n + 1]).
enum { FUSE_NONE = 0, FUSE_SILU = 1, FUSE_GELU = 2, FUSE_RELU2 = 3, FUSE_SWIGLU = 5 };

template <int FUSE>
__device__ __forceinline__ float gated_act(float x, float y, float alpha, float limit) {
  if constexpr (FUSE == FUSE_SILU) {
    return (x / (1.0f + expf(-x))) * y;
  } else if constexpr (FUSE == FUSE_SWIGLU) {
    const float gate = fminf(x, limit), up = fmaxf(-limit, fminf(y, limit));
    return gate * (1.0f / (1.0f + expf(-(gate * alpha)))) * (up + 1.0f);
  } else {
    const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return (x * (0.5f * (1.0f + tanhf(inner)))) * y;
  }
}

template <typename T, int MT, int NW, int FUSE, int WV = 4>  // WV waves per workgroup (8: the prefill tile, 128 x 256)
__global__ __launch_bounds__(64 * WV) void moe_bf16_kernel(T* __restrict__ out, const T* __restrict__ act,
                                                       const T* __restrict__ w, const float* __restrict__ bias,
                                                       const int32_t* __restrict__ rows_per_expert, int E, int N, int K,
                                                       int64_t ldb, int64_t w_stride_e, float act_alpha, float act_limit) {
  constexpr int BM = 16 * MT;
  constexpr int BN = 16 * NW * WV;
  constexpr int NT_ = 64 * WV;       // threads
  constexpr int AL = MT * 4 / WV;    // 16-byte activation chunks per thread and block
  static_assert(MT * 4 % WV == 0, "the chunks of a block must divide over the threads");
  __shared__ __attribute__((aligned(256))) char smem[2 * BM * 256];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;

  // ---- which (expert, block of its rows, column block): see moe_tiles.h
  constexpr bool kGated = FUSE == FUSE_SILU || FUSE == FUSE_GELU;
  // gpt-oss: gate / up are weight rows 2 n, 2 n + 1 - on the PLAIN tile layout (16 consecutive rows per wave tile: a pair sits in
  // neighbouring lanes, the epilogue fetches the partner's accumulator from lane l15 ^ 1; see moe_w4a16.hip)
  constexpr bool kPairs = FUSE == FUSE_SWIGLU;
  static_assert(!kGated || NW == 2, "the gated epilogue pairs the two n tiles of a wave");
  const int Nh = N >> 1;  // gated: output width; gate rows [0, Nh), up rows [Nh, N)
  const MoeTile tile = find_moe_tile(rows_per_expert, E, BM, kGated ? (Nh + BN / 2 - 1) / (BN / 2) : (N + BN - 1) / BN);
  if (tile.expert < 0) return;
  const int e = tile.expert, m0 = tile.m0, m_valid = tile.m_valid;
  const int n_base = kGated ? tile.col_block * (BN / 2) + wave * 16 : tile.col_block * BN + wave * (NW * 16);

  const T* wexp = w + (int64_t)e * w_stride_e;
  uint32_t woff[NW];  // element offset of this lane's row and k group
#pragma unroll
  for (int nt = 0; nt < NW; ++nt) {
    int n = kGated ? n_base + l15 + nt * Nh : n_base + nt * 16 + l15;
    const int lim = kGated ? (nt + 1) * Nh : N;
    n = n < lim ? n : lim - 1;
    woff[nt] = (uint32_t)n * (uint32_t)ldb + 8 * g;
  }

  // (a scalar base per workgroup and 32-bit per-thread offsets; the zeroing of chunks past K happens on the way to LDS and
  // not behind the load: a select right after a load waits for it on the spot, with every younger load in flight)
  const T* act_blk = act + (int64_t)m0 * K;
  uint32_t aoff[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int idx = i * NT_ + tid;
    const int row = idx >> 4, c = idx & 15;
    aoff[i] = (uint32_t)(row < m_valid ? row : m_valid - 1) * (uint32_t)K + c * 8;
  }
  auto load_a = [&](int kb, v4i (&r)[AL]) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int c = (i * NT_ + tid) & 15;
      const bool in = kb * 128 + c * 8 < K;  // past K: read k = 0 instead (zeroed in store_a; no branch)
      r[i] = *reinterpret_cast<const v4i*>(act_blk + (aoff[i] + (in ? (uint32_t)kb * 128u : 0u)));
    }
  };
  auto store_a = [&](int buf, int kb, const v4i (&r)[AL]) {
    char* base = smem + buf * (BM * 256);
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int idx = i * NT_ + tid;
      const int row = idx >> 4, c = idx & 15;
      const bool in = kb * 128 + c * 8 < K;
      const v4i zero = {0, 0, 0, 0};
      *reinterpret_cast<v4i*>(base + row * 256 + ((c ^ (row & 15)) << 4)) = in ? r[i] : zero;
    }
  };
  // weights of k step j of block kb: 16 bytes at k = 128 kb + 32 j + 8 g (past K: any valid address, the
  // activations there are zero)
  auto load_w = [&](int kb, v4i (&dst)[NW][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb * 128 + 32 * j;
      const uint32_t koff = (k + 8 * g < K) ? (uint32_t)k : 0u;
#pragma unroll
      for (int nt = 0; nt < NW; ++nt) dst[nt][j] = *reinterpret_cast<const v4i*>(wexp + woff[nt] + koff);
    }
  };

  v4f acc[MT][NW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  const int nkb = (K + 127) >> 7;
  v4i wq_[2][NW][4], aq_[2][AL];
  // (activation requests in front of the weight ring's, as in the steady state: the waits the compiler counts for the loop
  // are the worse of the two ways into it)
  // (and fenced: the scheduler otherwise interleaves the two slots' requests, slot 0 then looks as young as slot 1)
  load_a(0, aq_[0]);
  load_a(1, aq_[1]);
  store_a(0, 0, aq_[0]);
  load_a(2, aq_[0]);
  __builtin_amdgcn_sched_barrier(0);
  load_w(0, wq_[0]);
  __builtin_amdgcn_sched_barrier(0);
  load_w(1, wq_[1]);
  __builtin_amdgcn_sched_barrier(0);

  for (int kb0 = 0; kb0 < nkb; kb0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kb = kb0 + u;
      const int buf = kb & 1;
      // (not __syncthreads(): that would wait vmcnt(0) and drain the prefetch rings every block)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      store_a(buf ^ 1, kb + 1, aq_[(u + 1) & 1]);
      load_a(kb + 3, aq_[(u + 1) & 1]);
      const char* abase = smem + buf * (BM * 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = mt * 16 + l15;
          const v4i af = *reinterpret_cast<const v4i*>(abase + row * 256 + (((4 * j + g) ^ l15) << 4));
#pragma unroll
          for (int nt = 0; nt < NW; ++nt) acc[mt][nt] = mma16<T>(af, wq_[u][nt][j], acc[mt][nt]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // refill slot u AFTER its last use: requested earlier, the new block has to live in other registers and the loop end
      // moves the ring back into place with copies, each of which waits for the load into its source (the whole ring)
      load_w(kb + 2, wq_[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: lane owns out[m = 16 mt + 4 g + r][n = n_base + 16 nt + l15]
  if constexpr (kGated) {
    const int n = n_base + l15;
    if (n < Nh) {
      const float bg = bias ? bias[(int64_t)e * N + n] : 0.f, bu = bias ? bias[(int64_t)e * N + Nh + n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mt * 16 + 4 * g + r;
          if (row < m_valid)
            out[(int64_t)(m0 + row) * Nh + n] = (T)gated_act<FUSE>(acc[mt][0][r] + bg, acc[mt][1][r] + bu, act_alpha, act_limit);
        }
      }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      const int n = n_base + nt * 16 + l15;
      if (n >= N) continue;
      const float bv = bias ? bias[(int64_t)e * N + n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mt * 16 + 4 * g + r;
          float v = acc[mt][nt][r] + bv;
          if constexpr (kPairs) {  // even lane: gate (row n), its right neighbour: up (row n + 1); N even: both or neither in range
            const float other = __shfl_xor(v, 1, 64);
            if (!(l15 & 1) && row < m_valid) out[(int64_t)(m0 + row) * Nh + (n >> 1)] = (T)gated_act<FUSE>(v, other, act_alpha, act_limit);
            continue;
          }
          if constexpr (FUSE == FUSE_RELU2) {
            v = fmaxf(v, 0.f);
            v = v * v;
          }
          if (row < m_valid) out[(int64_t)(m0 + row) * N + n] = (T)v;
        }
      }
    }
  }
}

template <typename T, int MT, int NW, int WV = 4>
static int launch(hipStream_t st, void* out, const void* act, const void* w, const float* bias, const int32_t* rows,
                  int64_t total_m, int E, int N, int K, int64_t ldb, int64_t w_stride_e, int fuse) {
  constexpr int BM = 16 * MT, BN = 16 * NW * WV;
  const bool gated = fuse == FUSE_SILU || fuse == FUSE_GELU;  // (the gpt-oss swiglu runs on the plain layout)
  const int64_t wgs = moe_tile_launch_size(total_m, E, BM, gated ? cdiv(N / 2, BN / 2) : cdiv(N, BN));
  if (wgs >= ((int64_t)1 << 31)) return fail(SGLK_EINVAL, "moe_grouped_mm_nt_xe20: problem too large for one launch");
  dim3 grid((unsigned)wgs);
#define SGLK_GO(F) \
  moe_bf16_kernel<T, MT, NW, F, WV><<<grid, 64 * WV, 0, st>>>((T*)out, (const T*)act, (const T*)w, bias, rows, E | t_tail_flag, N, K, ldb, w_stride_e, t_act_alpha, t_act_limit)
  switch (fuse) {
    case FUSE_SILU: SGLK_GO(FUSE_SILU); break;
    case FUSE_GELU: SGLK_GO(FUSE_GELU); break;
    case FUSE_RELU2: SGLK_GO(FUSE_RELU2); break;
    case FUSE_SWIGLU: SGLK_GO(FUSE_SWIGLU); break;
    default: SGLK_GO(FUSE_NONE); break;
  }
#undef SGLK_GO
  return check_launch("moe_grouped_mm_nt_xe20");
}

#ifdef SGLK_PROBES
static int g_bf16_wv = 8;
#else
constexpr int g_bf16_wv = 8;
#endif

template <typename T>
static int dispatch(hipStream_t st, void* out, const void* act, const void* w, const float* bias, const int32_t* rows,
                    int64_t total_m, int E, int N, int K, int64_t ldb, int64_t w_stride_e, int fuse) {
  // tile policy by average rows per expert, as the reference (GroupGemmXe20.cpp:226-274); tail mode: total_m is the worst case
  // that sizes the launch, the tails of routed experts are a few dozen rows
  // (round 5, late - a token sweep with routed counts, Mixtral shapes, 16-bit weights: 554 us at 64 tokens, 721 at 96, 570 at 128; 606
  //  at 192, 828 at 256. A second row block of an expert streams its weights again, and with an average of 24 rows on 32-row tiles
  //  two or three of eight experts had one. The tile now holds the average plus three standard deviations of a routed count
  //  (avg + 3 sqrt(avg)) up to 32 rows; the 128-row tile's eight-wave kernel streams slower (~750 us against ~560), so the 64-row tile
  //  stays until about a quarter of the experts overflow it (average 60: 606 us at 192 tokens, 828 -> 759 at 256). A tail launch takes
  //  the 64-row tile: one pass behind 128-row blocks, at most two behind 256-row blocks (the 32-row tile took up to four passes over
  //  an expert's weights at 1536 tokens: 2018 us; the 128-row tile cost the short tails of 1024 tokens 180 us).)
  const int64_t avg = t_tail_flag ? 43 : total_m / E;
  if (avg <= 7) return launch<T, 1, 2>(st, out, act, w, bias, rows, total_m, E, N, K, ldb, w_stride_e, fuse);
  if (avg <= 18) return launch<T, 2, 2>(st, out, act, w, bias, rows, total_m, E, N, K, ldb, w_stride_e, fuse);
  if (avg <= 60) return launch<T, 4, 2>(st, out, act, w, bias, rows, total_m, E, N, K, ldb, w_stride_e, fuse);
  // (eight waves share the staged activation tile, 128 x 256: half the activation traffic and barriers per flop)
  // (61 .. 96 rows on average - below the tile pipeline's 88 or where it refuses the shape: the 128-row tile on four waves streams
  //  faster than on eight, 256 / 288 / 320 tokens 742 / 754 / 755 -> 678 / 676 / 694 us, lease zn)
  // (diagnostic build: 4 forces four waves, 9 eight, 8 = this rule)
  if (g_bf16_wv == 9 || (g_bf16_wv == 8 && avg > 96)) return launch<T, 8, 2, 8>(st, out, act, w, bias, rows, total_m, E, N, K, ldb, w_stride_e, fuse);
  return launch<T, 8, 2>(st, out, act, w, bias, rows, total_m, E, N, K, ldb, w_stride_e, fuse);
}

}  // namespace
}  // namespace sglk

static int moe_bf16_run(sglk_stream_t stream, void* out, const void* activations, const void* weights, const float* bias,
                        const int32_t* rows_per_expert, int64_t total_m, int64_t n_experts, int64_t N, int64_t K, int64_t ldb,
                        int64_t weight_stride_e, int dtype, int fused_act, float act_alpha, float act_limit);

extern "C" int sglk_moe_grouped_mm(sglk_stream_t stream, void* out, const void* activations, const void* weights,
                                   const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                   int64_t n_experts, int64_t N, int64_t K, int64_t ldb, int64_t weight_stride_e,
                                   int dtype, int fused_act) {
  using namespace sglk;
  SGLK_REQUIRE(fused_act >= 0 && fused_act <= 3, "moe_grouped_mm_nt_xe20: fused_act must be 0 (none), 1 (silu), 2 (gelu) or 3 (relu2)");
  return moe_bf16_run(stream, out, activations, weights, bias, rows_per_expert, total_m, n_experts, N, K, ldb, weight_stride_e, dtype,
                      fused_act, 0.f, 0.f);
}

extern "C" int sglk_moe_grouped_mm_swiglu(sglk_stream_t stream, void* out, const void* activations, const void* weights,
                                          const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                          int64_t n_experts, int64_t N, int64_t K, int64_t ldb, int64_t weight_stride_e,
                                          int dtype, float alpha, float limit) {
  using namespace sglk;
  SGLK_REQUIRE(limit > 0.f, "moe_grouped_mm_nt_xe20: the gpt-oss swiglu needs a positive gemm1_limit");
  return moe_bf16_run(stream, out, activations, weights, bias, rows_per_expert, total_m, n_experts, N, K, ldb, weight_stride_e, dtype,
                      5, alpha, limit);
}

static int moe_bf16_run(sglk_stream_t stream, void* out, const void* activations, const void* weights, const float* bias,
                        const int32_t* rows_per_expert, int64_t total_m, int64_t n_experts, int64_t N, int64_t K, int64_t ldb,
                        int64_t weight_stride_e, int dtype, int fused_act, float act_alpha, float act_limit) {
  using namespace sglk;
  t_act_alpha = act_alpha;
  t_act_limit = act_limit;
  SGLK_REQUIRE(dtype == SGLK_BF16 || dtype == SGLK_F16, "moe_grouped_mm_nt_xe20: activations and weights must be bfloat16 or half");
  SGLK_REQUIRE(n_experts > 0 && N > 0 && K > 0, "moe_grouped_mm_nt_xe20: bad shape");
  SGLK_REQUIRE(K % 8 == 0 && ldb % 8 == 0 && weight_stride_e % 8 == 0, "moe_grouped_mm_nt_xe20: K and the weight strides must be multiples of 8");
  SGLK_REQUIRE((uintptr_t)activations % 16 == 0 && (uintptr_t)weights % 16 == 0,
               "moe_grouped_mm_nt_xe20: activations and weights must be 16-byte aligned");
  SGLK_REQUIRE(N * ldb < (1ll << 32), "moe_grouped_mm_nt_xe20: one expert's weights must stay below 4 Gi elements");
  SGLK_REQUIRE(!(fused_act == 1 || fused_act == 2 || fused_act == 5) || N % 2 == 0, "moe_grouped_mm_nt_xe20: a gated epilogue needs an even N");
  if (total_m == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = moe_persist_try(st, out, activations, weights, nullptr, nullptr, 0, bias, rows_per_expert, total_m, (int)n_experts, (int)N,
                               (int)K, ldb, weight_stride_e, dtype, 0, fused_act, act_limit, act_alpha)) {
    if (rc < 0) return rc;
    if (rc == 3) return SGLK_OK;  // (the tile pipeline took the remainders too)
    t_tail_flag = rc == 2 ? kMoeTailFlag128 : kMoeTailFlag;  // the experts' last rows (at most 128 each) on the streaming kernel
    const int64_t tail_m = std::min<int64_t>(total_m, (rc == 2 ? 64 : 128) * n_experts);
    rc = dtype == SGLK_BF16 ? dispatch<bf16>(st, out, activations, weights, bias, rows_per_expert, tail_m, (int)n_experts, (int)N,
                                             (int)K, ldb, weight_stride_e, fused_act)
                            : dispatch<f16>(st, out, activations, weights, bias, rows_per_expert, tail_m, (int)n_experts, (int)N,
                                            (int)K, ldb, weight_stride_e, fused_act);
    t_tail_flag = 0;
    return rc;
  }
  if (dtype == SGLK_BF16)
    return dispatch<bf16>(st, out, activations, weights, bias, rows_per_expert, total_m, (int)n_experts, (int)N, (int)K, ldb,
                          weight_stride_e, fused_act);
  return dispatch<f16>(st, out, activations, weights, bias, rows_per_expert, total_m, (int)n_experts, (int)N, (int)K, ldb,
                       weight_stride_e, fused_act);
}

#ifdef SGLK_PROBES
extern "C" SGLK_API void sglk_debug_set_moe_bf16_waves(int wv) { sglk::g_bf16_wv = wv; }
#endif
