// W4A16 (int4 weights, bf16/fp16 activations) grouped GEMM for MoE experts on gfx950.
//
// Replaces reference src/sycl/GroupGemmW4A16Xe20.cpp:92-283 (host) and
// src/sycl/kernels/moe/xe20/w4a16/* (CuTe kernel). Contract kept (schema
// src/torch_extension_sycl.cc:214-217, the `_xe20` name included because python/sgl_kernel/moe.py:751
// calls it by that name):
//   for expert e with rows[e] consecutive rows of `activations`:
//     out_e = A_e @ dq(W_e)^T (+ bias_e),   dq = (code - zp) * scale
//   W [E, N, K/2] bytes, low nibble = even k; codes are two's-complement when zeros is absent,
//   unsigned otherwise; scales / zeros [E, N, K/group] in the activation dtype; bias fp32 [E, N].
//
// Arithmetic. The reference dequantises every weight to the activation dtype (one rounding of
// (code - zp) * scale, gemm_xe2.hpp:52-76, :405-428) and feeds bf16 x bf16 DPAS. Here the 4-bit codes
// stay EXACT all the way through the matrix core and the scale is applied in fp32 per quantisation group:
//   out[m,n] = sum_groups scale[n,g] * ( sum_{k in g} (16 + u[n,k]) * a[m,k]  -  (16 + zp[n,g]) * sum_{k in g} a[m,k] )
// where 16 + u is formed in the 16-bit float format by pure bit operations (0x4180 | u<<3 is the bf16
// 16 + u; 0x4C00 | u<<6 the fp16 one) and sum_k a[m,k] comes from one extra MFMA against a fragment
// of ones. That is 8 bit-ops per 8 weights instead of ~28 VALU ops for convert/subtract/multiply/round,
// which is what keeps the decode regime on the HBM roofline, and it is at least as accurate as the
// reference (no per-weight rounding); results agree within the reference test tolerance.
//
// mxfp4 weights (FMT = 1; reference GroupGemmW4A16Xe20.cpp:140-168, gemm_xe2.hpp:238-448): the nibbles are OCP e2m1
// codes (sign, 2 exponent bits, 1 mantissa bit: 0, 0.5, 1, 1.5, 2, 3, 4, 6), one E8M0 scale byte per 32 weights.
// Every e2m1 value is exact in bf16 / fp16: the codes are widened by two byte-table lookups (v_perm_b32 on the 3-bit
// magnitudes of a nibble pair) plus the sign bits, keeping the nibble-pair order of the int4 path; the group scale
// 2^(byte - 127) is applied in fp32 per 32-deep k step (no zero point, no row-sum MFMA).
//
// Data movement. Weights are streamed once from HBM straight into registers (16 B per lane; they are not
// shared between waves, so an LDS round trip would be pure overhead). For quantisation groups of 32 / 64 a 4x4
// dword transpose across the four 16-lane groups (2 x v_permlane32_swap + 2 x v_permlane16_swap) gives every MFMA
// k-step a contiguous 32-wide k range, i.e. exactly one group; groups of 128 / 256 cover the whole 128-deep block,
// so the k order inside it is permuted instead (k-step j = dword j of each lane) and nothing moves between lanes. The
// activation tile [BM rows x 128 k] is staged through LDS once per 128-deep block for all 4 waves
// (element order permuted to match the nibble-pair order of the weight fragments; rows XOR-swizzled).
// Block = 4 waves; wave w owns NW 16-wide n tiles and all MT 16-row m tiles of the block's expert rows.
// Experts are ragged: the grid is sized for the worst case and each block finds its (expert, row block)
// from rows_per_expert on the device (no host sync).
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "moe_tiles.h"

namespace sglk {
// moe_persist.hip: the dense tile pipeline for prefill row counts (returns 1 after launching, 0 if the shape does not qualify)
int moe_persist_try(hipStream_t st, void* out, const void* act, const void* w, const void* scales, const void* zeros,
                    int group_shift, const float* bias,
                    const int32_t* rows, int64_t total_m, int E, int N, int K, int64_t ldb, int64_t stride_e, int dtype, int w4,
                    int fuse, float act_limit, float act_alpha = 0.f);
// moe_persist.hip: the K split of the 128-row blocks into fp32 slabs (returns 2 after launching, 0 if it does not apply)
int moe_persist_splitk_try(hipStream_t st, float* ws, const void* act, const void* w, const void* scales, const void* zeros,
                           int group_shift, const int32_t* rows, int64_t total_m, int E, int N, int K, int dtype, int w4);
int moe_persist_splitk_applies(int64_t total_m, int E, int N, int K, int group_shift, int w4, int dtype);
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <typename T>
struct W4;
template <>
struct W4<bf16> {
  static constexpr int kShift = 3;                 // nibble position inside a 16-bit half: 16 + u = 0x4180 | u << 3
  static constexpr uint32_t kMagic = 0x41804180u;
  static constexpr uint32_t kOnes = 0x3F803F80u;   // (1.0, 1.0)
  // e2m1 magnitudes 0, .5, 1, 1.5 | 2, 3, 4, 6 as bf16: high bytes 00 3F 3F 3F | 40 40 40 40, low bytes 00 00 80 C0 | 00 40 80 C0
  static constexpr uint32_t kHiLo = 0x3F3F3F00u, kHiHi = 0x40404040u, kLoLo = 0xC0800000u, kLoHi = 0xC0804000u;
  static __device__ __forceinline__ v4f mma(const v4i& a, const v4i& b, const v4f& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
  }
};
template <>
struct W4<f16> {
  static constexpr int kShift = 6;                 // 16 + u = 0x4C00 | u << 6
  static constexpr uint32_t kMagic = 0x4C004C00u;
  static constexpr uint32_t kOnes = 0x3C003C00u;
  // as fp16: 0000 3800 3C00 3E00 | 4000 4200 4400 4600
  static constexpr uint32_t kHiLo = 0x3E3C3800u, kHiHi = 0x46444240u, kLoLo = 0u, kLoHi = 0u;
  static __device__ __forceinline__ v4f mma(const v4i& a, const v4i& b, const v4f& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
};

// 8 nibbles (k offsets 0..7, nibble i at bits 4i) -> 4 dwords of two 16-bit floats (16 + u):
// dword p = (k offset p, k offset p + 4).  The activation fragments use the same element order.
// Two instructions per dword: a shift and v_and_or_b32. The mask and the magic are passed as opaque register values:
// with literal operands the compiler emits and + or (a VOP3 takes no literal on gfx9). Not inline asm: the hazard
// recogniser does not see through an asm statement, and a VALU write it cannot see may land on the source registers of
// an MFMA still in the queue (that broke the fp16 / 32-row / group-64 instantiation).
template <typename T>
__device__ __forceinline__ v4i expand_nibbles(uint32_t w, uint32_t mask_reg, uint32_t magic_reg) {
  constexpr int S = W4<T>::kShift;
  v4i r;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int sh = S - 4 * p;
    const uint32_t t = sh >= 0 ? (w << sh) : (w >> (-sh));
    r[p] = (int)((t & mask_reg) | magic_reg);
  }
  return r;
}

// mxfp4: 8 e2m1 nibbles -> 4 dwords of two 16-bit floats in the same order (dword p = k offsets p, p + 4)
template <typename T>
__device__ __forceinline__ v4i expand_mxfp4(uint32_t w) {
  v4i r;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t uu = w >> (4 * p);
    const uint32_t mag = uu & 0x00070007u;      // magnitudes of nibbles p (byte 0) and p + 4 (byte 2)
    const uint32_t sel = mag | (mag << 8);      // same index in bytes (0, 1) and (2, 3)
    const uint32_t hi = __builtin_amdgcn_perm(W4<T>::kHiHi, W4<T>::kHiLo, sel);
    uint32_t v = hi & 0xFF00FF00u;
    if constexpr (W4<T>::kLoLo != 0u || W4<T>::kLoHi != 0u) {
      const uint32_t lo = __builtin_amdgcn_perm(W4<T>::kLoHi, W4<T>::kLoLo, sel);
      v |= lo & 0x00FF00FFu;
    }
    r[p] = (int)(v | ((uu & 0x00080008u) << 12));  // sign bits 3 / 19 -> 15 / 31
  }
  return r;
}

// mxfp4 through the conversion instruction of gfx950: v_cvt_scalef32_pk_bf16_fp4 turns the two e2m1 codes of one byte
// (low nibble first) into two bf16 times 2^(exponent of the f32 operand) - exact, scale included, one instruction per
// dword of results (tools/fp4_cvt_probe.cpp prints what it does at the ends of the E8M0 range: 2^-127 comes out as the
// bf16 subnormal, the top of the range as inf). dword p = k offsets 2p, 2p + 1: the natural order.
__device__ __forceinline__ v4i expand_mxfp4_hw(uint32_t w, float scale) {
  typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
  v4i r;
  r[0] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 0));
  r[1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 1));
  r[2] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 2));
  r[3] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 3));
  return r;
}

// 4x4 transpose of dwords across the four 16-lane groups: in: lane group g holds d[t] = M[g][t];
// out: lane group g holds d[t] = M[t][g].
__device__ __forceinline__ void transpose4(uint32_t (&d)[4]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    auto r = __builtin_amdgcn_permlane32_swap(d[t], d[t + 2], false, false);
    d[t] = r[0];
    d[t + 2] = r[1];
  }
#pragma unroll
  for (int t = 0; t < 4; t += 2) {
    auto r = __builtin_amdgcn_permlane16_swap(d[t], d[t + 1], false, false);
    d[t] = r[0];
    d[t + 1] = r[1];
  }
}

template <int I>
struct IntC {
  static constexpr int value = I;
};
template <typename F>
__device__ __forceinline__ void static_for4(F&& f) {
  f(IntC<0>{});
  f(IntC<1>{});
  f(IntC<2>{});
  f(IntC<3>{});
}

// PB: scale groups per 128-deep block (4, 2, 1 for groups 32, 64, >= 128); FMT: 0 = int4, two's-complement codes (scales
// of type T), 2 = int4, unsigned codes with zero points of type T, 1 = mxfp4 (scales = E8M0 bytes, group 32, no zeros),
// 3 = mxfp4 with bf16 activations through v_cvt_scalef32_pk_bf16_fp4: the 16 bytes a lane loads per 128-deep block are ONE
// scale group (k = 32 g ..), the conversion applies that lane's scale, so the block needs no per-group fold, no lane
// transposes, and runs on the PB = 1 machinery (scales read as one dword = the block's four E8M0 bytes per row)
template <typename T, int MT, int NW, int PB, int FMT, int WV = 4>  // WV waves per workgroup (8: the prefill tile, 64 x 256)
__global__ __launch_bounds__(64 * WV, (MT <= 2 ? 2 : 1)) void moe_w4a16_kernel(T* __restrict__ out, const T* __restrict__ act,
                                                        const uint8_t* __restrict__ wq, const void* __restrict__ scales_,
                                                        const void* __restrict__ zeros_, const float* __restrict__ bias,
                                                        const int32_t* __restrict__ rows_per_expert, int E, int N,
                                                        int K, int group_shift, int probe, int fuse, float act_limit,
                                                        const int32_t* __restrict__ row_map, float act_alpha) {
  // row_map (may be null): activation row of expert-contiguous row r is act[row_map[r]] - the token gather of fused_experts
  // (reference shuffle_rows, python/sgl_kernel/moe.py:739) folded into this kernel's staging loads: no [rows, K] copy of the
  // tokens, one launch less per call at decode sizes.
  // fuse (the gate / up activation of fused_experts in this GEMM's epilogue, as moe_bf16.hip does for 16-bit weights;
  // reference python/sgl_kernel/moe.py:751-835 runs GEMM, then a separate act-and-mul over a [rows, 2I] intermediate):
  //   1 silu, 2 gelu (tanh form): W holds gate rows [0, N/2) then up rows [N/2, N); a workgroup takes BN/2 gate columns
  //   AND the BN/2 up columns that go with them (wave tiles nt < NW/2 gate, nt >= NW/2 up), out[m, n] = T(act(gate) * up)
  //   is [total_m, N/2]: the product is formed on the fp32 accumulators, one rounding. 3 relu2: out = T(max(x, 0)^2).
  //   4 DeepSeek-V4 swiglu (reference silu_and_mul_clamp, python/sgl_kernel/elementwise.py:231-255): gate = min(gate, limit),
  //   up = clamp(up, -limit, limit), then silu(gate) * up.
  // probe (libsglk_probes.so only; 0 in the release library): timing experiments with garbage results -
  // 1: one activation row for all 16 m rows, 2: no output stores, 4: non-temporal weight loads, 8: scales read once,
  // 16: no weight expansion / MFMAs (stream only), 32: no barrier, 64: no activation staging, 512: row blocks fastest in the
  // tile order also for the 64-row tiles. (Probes inside the MFMA steps - no MFMAs, no expansion - were used once and removed:
  // their branches cost the diagnostic build 35 % at prefill sizes.)
#ifndef SGLK_PROBES
  probe = 0;
#endif
  constexpr int BM = 16 * MT;
  constexpr int BN = 16 * NW * WV;
  constexpr int NT_ = 64 * WV;  // threads
  // kD: 128-deep blocks of weights / scales in flight per wave (register ring, the K loop is unrolled kD times).
  // (decode tiles: 8 KiB of weights in flight per wave; the large tiles: registers. A spill in this loop is reloaded
  // through scratch, i.e. behind an s_waitcnt vmcnt(0) that empties the rings: build.py's check_isa rejects one.)
  constexpr int kD = (PB == 1 && (MT == 1 || (MT == 2 && FMT != 2))) ? (NW == 1 ? 8 : 4) : 2;
  // AS: 128-deep blocks per activation stage. The decode tiles stage 512 / 256 k at a time: one workgroup barrier per
  // four / two blocks instead of one per block (with one workgroup per CU - the Mixtral down projection at decode - barrier and
  // staging cost 40 of 96 us).
  constexpr int AS = kD >= 4 ? (MT == 1 ? 4 : 2) : 1;  // (32-row tiles: 256 k per stage, four more would spill)
  constexpr int AROW = 256 * AS;  // bytes of a staged activation row
  constexpr bool kTranspose = PB > 1;  // (groups of 32 / 64: see below, where the weights are expanded)
  // SV: the scales (zero points) of the kD = 4 / 8 blocks of one trip of the K loop are ONE 8- / 16-byte load per row,
  // requested a trip ahead (groups of 128, K a multiple of 512 / 1024: the host sends other shapes elsewhere). A quarter of the scale
  // requests, and the one loop-carried register set is rotated at the top of the trip, where its load is the oldest in flight.
  constexpr bool SV = PB == 1 && kD >= 4 && FMT != 1;  // (FMT 3: the kD dwords of E8M0 bytes of a trip, one 16- / 32-byte load)
  __shared__ __attribute__((aligned(256))) char smem[2 * BM * AROW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;

  // ---- which (expert, block of its rows, column block): see moe_tiles.h
  // (64-row tiles of a projection with more k than columns: the column blocks of a row block run together and share its
  // activations in L2 - 731 against 755 us for the Mixtral down projection at 512 rows per expert; the gate / up projection
  // measured the other way round, 1609 against 1706 us. Probe 512 flips the choice.)
  // fuse 5: the gpt-oss swiglu (reference kernels/moe/xe20/common/activation.hpp:31-42, moe_kernel.hpp:109-125 - the fused form of
  // the reference's 16-bit GEMM): gate = weight row 2 n, up = row 2 n + 1 (INTERLEAVED), bias likewise. Here it runs on the PLAIN
  // tile layout - a wave tile is 16 consecutive weight rows, i.e. eight (gate, up) pairs in neighbouring lanes - and the epilogue
  // fetches the partner's accumulator from lane l15 ^ 1. (First form: the gated layout with the row map 2 n / 2 n + 1 - a load
  // instruction then touches 16 rows 4 KiB apart: 172 us against 132 for the split-halves silu at 64 Mixtral tokens, and the
  // decode-sized tile forms of the plain GEMM - 64-column workgroups, 16-column wave tiles - were out of reach.)
  const bool gated = NW >= 2 && (fuse == 1 || fuse == 2 || fuse == 4);
  const bool pairs = fuse == 5;
  const int Nh = N >> 1;  // gated: output width
  const int col_blocks = gated ? (Nh + BN / 2 - 1) / (BN / 2) : (N + BN - 1) / BN;
  const MoeTile tile = find_moe_tile(rows_per_expert, E, BM, col_blocks, MT >= 4 && ((N < K) != ((probe & 512) != 0)));
  if (tile.expert < 0) return;
  const int e = tile.expert, m0 = tile.m0, m_valid = tile.m_valid;
  const int n_base = tile.col_block * BN + wave * (NW * 16);
  // weight row of this lane for wave tile nt (gated: the up half mirrors the gate half)
  auto col_of = [&](int nt) -> int {
    if (!gated) return n_base + nt * 16 + l15;
    constexpr int H2 = NW >= 2 ? NW / 2 : 1;
    const int ng = tile.col_block * (BN / 2) + wave * (H2 * 16) + (nt % H2) * 16 + l15;  // gate column = output column
    return nt < H2 ? ng : Nh + ng;
  };
  auto out_col_of = [&](int nt) -> int {  // output column of gate tile nt (nt < NW / 2)
    constexpr int H2 = NW >= 2 ? NW / 2 : 1;
    return tile.col_block * (BN / 2) + wave * (H2 * 16) + (nt % H2) * 16 + l15;
  };
  const int n_lim = gated ? Nh : N;  // valid gate / plain columns

  using S = typename std::conditional<FMT == 1, uint8_t, typename std::conditional<FMT == 3, uint32_t, T>::type>::type;  // stored scale type
  static_assert(FMT != 3 || (PB == 1 && std::is_same<T, bf16>::value), "the fp4 conversion path: bf16, one dword of scales per block");
  const S* scales = reinterpret_cast<const S*>(scales_);
  const S* zeros = reinterpret_cast<const S*>(zeros_);
  const int kgroups = K >> group_shift;         // scales per row
  constexpr bool has_zp = FMT == 2;
  constexpr bool is_int4 = FMT == 0 || FMT == 2;

  // ---- per-lane weight / scale rows (clamped; stores are guarded): 32-bit offsets from per-expert bases
  const uint8_t* wexp = wq + (int64_t)e * N * (K / 2);
  const S* sexp = scales + (int64_t)e * N * kgroups;
  const S* zexp = has_zp ? zeros + (int64_t)e * N * kgroups : sexp;
  uint32_t woff[NW], soff[NW];
#pragma unroll
  for (int nt = 0; nt < NW; ++nt) {
    int n = col_of(nt);
    n = n < N ? n : N - 1;
    woff[nt] = (uint32_t)n * (uint32_t)(K / 2) + 16 * g;
    soff[nt] = (uint32_t)n * (uint32_t)kgroups;
  }

  // ---- activation staging: thread handles 16-byte chunks (row, c) of the [BM][128] tile. Split in a load (global ->
  // registers) and a store (registers -> LDS) one iteration later: a load consumed in the iteration that issues it
  // exposes a full memory latency per 128-deep block (that alone was 200 us of a K = 14336 decode GEMM).
  // (a scalar base per workgroup and 32-bit per-thread offsets: 64-bit per-thread row addresses of the large tiles were
  // spilled and reloaded inside the K loop)
  const T* act_blk = row_map ? act : act + (int64_t)m0 * K;
  constexpr int AL = MT * AS * 4 / WV;  // 16-byte chunks per thread and stage: chunk q = i * NT_ + tid of [BM][16 AS]
  static_assert(MT * AS * 4 % WV == 0, "stage chunks must divide over the threads");
  uint32_t aoff[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int q = i * NT_ + tid;
    const int row = q / (16 * AS), c = q % (16 * AS);
    const int rc = (probe & 1) ? 0 : row < m_valid ? row : m_valid - 1;
    aoff[i] = (uint32_t)(row_map ? row_map[m0 + rc] : rc) * (uint32_t)K + c * 8;
  }
  auto load_a = [&](int st, v4i (&r)[AL]) {  // stage st = blocks st * AS .. + AS - 1
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int c = (i * NT_ + tid) % (16 * AS);
      const bool in = st * (128 * AS) + c * 8 < K;  // past K: read k = 0 instead (zeroed on the way to LDS; no branch)
      r[i] = *reinterpret_cast<const v4i*>(act_blk + (aoff[i] + (in ? (uint32_t)st * (128u * AS) : 0u)));
    }
  };
  // (the zeroing of chunks past K belongs here and not behind the load: a select right after the load waits for it on the
  // spot, with every younger load in flight)
  auto store_a = [&](int buf, int st, const v4i (&r)[AL]) {
    char* base = smem + buf * (BM * AROW);
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int q = i * NT_ + tid;
      const int row = q / (16 * AS), c = q % (16 * AS);
      const bool in = st * (128 * AS) + c * 8 < K;
      const v4i zero = {0, 0, 0, 0};
      const v4i v = in ? r[i] : zero;
      // element order (a0,a4,a1,a5,a2,a6,a3,a7) to match expand_nibbles
      v4i p = v;  // (FMT 3: the conversion delivers the natural element order)
      if constexpr (FMT != 3) {
        p[0] = (int)__builtin_amdgcn_perm((uint32_t)v[2], (uint32_t)v[0], 0x05040100u);
        p[1] = (int)__builtin_amdgcn_perm((uint32_t)v[2], (uint32_t)v[0], 0x07060302u);
        p[2] = (int)__builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[1], 0x05040100u);
        p[3] = (int)__builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[1], 0x07060302u);
      }
      // (the swizzle permutes the 16 chunks of a 128-deep block among themselves)
      // position of chunk cb = c % 16 of a 128-deep block inside its 256-byte row piece: slot(cb) ^ (row & 15); slot is the
      // identity when the k-steps read chunks 4 j + g, and {0, 12, 4, 8}[cb / 4] + cb % 4 when they read chunks 4 g + j
      // (groups >= 128): a ds_read_b128 serves lanes {0-3, 12-15, 20-27} together (MI355X_MICROARCH.md, LDS), i.e. rows
      // {0-3, 12-15} of lane group 0 with rows 4-11 of lane group 1 - their slots must differ by a value with equal bits 2
      // and 3 (here 12) or every read is a 2-way conflict (3 conflict cycles per LDS instruction in the r02 profile of
      // the first version); the eight chunks a ds_write_b128 group stores stay distinct mod 8.
      const int cb = c & 15;
      const int slot = kTranspose ? cb : (((0x84C0 >> (cb & 12)) & 15) + (cb & 3));
      *reinterpret_cast<v4i*>(base + row * AROW + (((c & ~15) | (slot ^ (row & 15))) << 4)) = p;
    }
  };

  v4f acc[MT][NW], part[MT][NW], asum[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    asum[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
      part[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
  }
  const int nkb = (K + 127) >> 7;   // K is a multiple of 32; the last 128-block may hold 1..3 k steps
  const int ksteps = K >> 5;
  const v4i ones = {(int)W4<T>::kOnes, (int)W4<T>::kOnes, (int)W4<T>::kOnes, (int)W4<T>::kOnes};
  uint32_t magic = W4<T>::kMagic, nib_mask = (0xFu << W4<T>::kShift) | (0xFu << (W4<T>::kShift + 16));
  asm volatile("" : "+v"(magic), "+s"(nib_mask));  // (opaque register values: see expand_nibbles)
  // One scale group per 128-deep block (groups of 128 / 256): the order of k inside the block is free, so MFMA k-step j
  // takes dword j of every lane's own 16 weight bytes (k = 32 g + 8 j ..) and the activation fragment is read from
  // that k range; smaller groups need each k-step to be one contiguous 32-wide range: 4x4 transpose across lane groups.

  // weights run two 128-deep blocks ahead of the MFMAs, scales / zero points one block ahead: at decode sizes the
  // kernel is a latency-bound HBM stream and a load consumed in the iteration that issued it stalls every wave
  // All three streams (weights, scales / zero points, activations) run kD 128-deep blocks ahead of the MFMAs in
  // register rings with static slots (the K loop is unrolled kD times): at decode sizes an iteration is ~0.15 us of
  // MFMA work against ~2 us of memory latency, and a load consumed close to where it was issued stalls the wave.
  uint32_t wd[NW][4], wq_[kD][NW][4];
  // weights past K are never multiplied by anything but zero activations: any valid address will do (block 0)
  auto load_w = [&](int kb, uint32_t (&dst)[NW][4]) {
    const uint32_t koff = (kb * 128 + 32 * g < K) ? (uint32_t)kb * 64u : 0u;
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      const v4i* wp = reinterpret_cast<const v4i*>(wexp + woff[nt] - 16 * g + ((kb * 128 + 32 * g < K) ? 16 * g : 0) + koff);
      const v4i t = (probe & 4) ? __builtin_nontemporal_load(wp) : *wp;
      dst[nt][0] = t[0]; dst[nt][1] = t[1]; dst[nt][2] = t[2]; dst[nt][3] = t[3];
    }
  };
  S sq_[kD][NW][PB], zq_[has_zp ? kD : 1][NW][PB];
  auto load_s = [&](int kb, S (&sd)[NW][PB], S (&zd)[NW][PB]) {
    const int kg0 = (kb * 128) >> group_shift;
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        int kg = (probe & 8) ? 0 : kg0 + i;
        kg = kg < kgroups ? kg : kgroups - 1;
        sd[nt][i] = sexp[soff[nt] + kg];
        if constexpr (has_zp) zd[nt][i] = zexp[soff[nt] + kg];
      }
    }
  };
  // (kD 16-bit scales of a row = kD / 2 dwords: one 8- or 16-byte load)
  typedef uint32_t SVec __attribute__((ext_vector_type(kD >= 4 ? (FMT == 3 ? kD : kD / 2) : 2)));
  SVec sv_cur[NW], sv_nxt[NW], zv_cur[NW], zv_nxt[NW];
  auto load_sv = [&](int kb0, SVec (&sd)[NW], SVec (&zd)[NW]) {
    int kg0 = (probe & 8) ? 0 : kb0;
    kg0 = kg0 + kD <= kgroups ? kg0 : kgroups - kD;
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      sd[nt] = *reinterpret_cast<const SVec*>(sexp + soff[nt] + kg0);
      if constexpr (has_zp) zd[nt] = *reinterpret_cast<const SVec*>(zexp + soff[nt] + kg0);
    }
  };
  auto sv_get = [&](const SVec& v, int u) -> float {  // element u of the packed 16-bit floats, widened
    const uint32_t d = v[u >> 1];
    const uint16_t h = (u & 1) ? (uint16_t)(d >> 16) : (uint16_t)d;
    return (float)__builtin_bit_cast(T, h);
  };
  // activations: with AS = 1 a ring of kD blocks like the weights; with AS = 4 one register set, a stage is requested
  // when the one before it goes to LDS (four blocks of lead)
  constexpr int kA = AS == 1 ? kD : 1;
  v4i aq_[kA][AL];
  // (activation requests in front of the ring's, as in the steady state: the waits the compiler counts for the loop are the
  // worse of the two ways into it)
#pragma unroll
  for (int d = 0; d < kA; ++d) load_a(d, aq_[d]);
  store_a(0, 0, aq_[0]);
  load_a(kA, aq_[0]);  // slot 0 now carries block / stage kA (zeros past K)
  if constexpr (SV) {
    load_sv(0, sv_cur, zv_cur);
    load_sv(kD, sv_nxt, zv_nxt);
  }
  __builtin_amdgcn_sched_barrier(0);  // (fenced: interleaved by the scheduler, slot 0 looks as young as the last slot)
#pragma unroll
  for (int d = 0; d < kD; ++d) {
    load_w(d, wq_[d]);
    if constexpr (!SV) load_s(d, sq_[d], zq_[has_zp ? d : 0]);
    __builtin_amdgcn_sched_barrier(0);
  }

  // (blocks past K run on zero activations: the trip count is padded to a multiple of kD, the body has no exit)
  for (int kb0 = 0; kb0 < nkb; kb0 += kD) {
#pragma unroll
  for (int u = 0; u < kD; ++u) {
    const int kb = kb0 + u;
    const int st = kb / AS;  // activation stage of this block
    const int buf = st & 1;
    if constexpr (SV) {
      if (u == 0 && kb0 > 0) {
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) {
          sv_cur[nt] = sv_nxt[nt];
          if constexpr (has_zp) zv_cur[nt] = zv_nxt[nt];
        }
        load_sv(kb0 + kD, sv_nxt, zv_nxt);
      }
    }
    if (u % AS == 0) {
      // stage st is in LDS; everyone is done reading the other buffer. Not __syncthreads(): that waits vmcnt(0) and
      // would drain the prefetch rings every iteration (measured: 3.7 us per 128-deep block instead of ~0.4)
      if (!(probe & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // AS = 1: block kb+1 sits in slot (u+1) % kD (slot 0 after the wrap was refilled with block kb0 + kD): stage it,
      // then refill that slot with block kb + 1 + kD. AS = 4: the one set holds stage st + 1; then request st + 2.
      if (!(probe & 64)) {
        store_a(buf ^ 1, st + 1, aq_[AS == 1 ? (u + 1) % kD : 0]);
        load_a(st + 1 + kA, aq_[AS == 1 ? (u + 1) % kD : 0]);
      }
    }
    S sc[NW][PB], zc[NW][PB];
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wd[nt][t] = wq_[u][nt][t];
      if constexpr (!SV) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          sc[nt][i] = sq_[u][nt][i];
          if constexpr (has_zp) zc[nt][i] = zq_[u][nt][i];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      if constexpr (FMT == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) wd[nt][t] ^= 0x88888888u;  // two's complement -> offset binary (zp 8)
      }
      if constexpr (kTranspose) transpose4(wd[nt]);  // now wd[nt][j] = codes of k = 128 kb + 32 j + 8 g .. + 7 of row n
    }
    const char* abase = smem + buf * (BM * AROW) + (kb % AS) * 256;
    if (probe & 16) {
#pragma unroll
      for (int nt = 0; nt < NW; ++nt) acc[0][nt][0] += __uint_as_float(wd[nt][0] ^ wd[nt][1] ^ wd[nt][2] ^ wd[nt][3]) + (SV ? sv_get(sv_cur[nt], u) : (float)sc[nt][0]);
    } else {
    // (requesting the activation fragments of the whole block up front, ahead of the first step's expansion, measured no
    // faster on the decode tiles: 142 -> 147 us)
    float lane_scale[NW];  // FMT 3: 2^(E8M0 byte - 127) of this lane's group (k = 128 kb + 32 g ..); byte 0 = the f32 subnormal
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      lane_scale[nt] = 1.f;
      if constexpr (FMT == 3) {
        const uint32_t e8 = ((SV ? (uint32_t)sv_cur[nt][u] : (uint32_t)sc[nt][0]) >> (8 * g)) & 0xffu;
        lane_scale[nt] = __uint_as_float(e8 ? e8 << 23 : 0x00400000u);
      }
    }
    static_for4([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // (k steps past K multiply zero activations: no tail branch)
      v4i wf[NW];
#pragma unroll
      for (int nt = 0; nt < NW; ++nt) {
        if constexpr (FMT == 3) wf[nt] = expand_mxfp4_hw(wd[nt][j], lane_scale[nt]);
        else wf[nt] = !is_int4 ? expand_mxfp4<T>(wd[nt][j]) : expand_nibbles<T>(wd[nt][j], nib_mask, magic);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + l15;
        const int slot = kTranspose ? 4 * j + g : ((0x84C0 >> (4 * g)) & 15) + j;  // (see store_a)
        const v4i af = *reinterpret_cast<const v4i*>(abase + row * AROW + ((slot ^ l15) << 4));
        if constexpr (is_int4) asum[mt] = W4<T>::mma(af, ones, asum[mt]);
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) part[mt][nt] = W4<T>::mma(af, wf[nt], part[mt][nt]);
      }
      // end of a scale segment (compile-time position: every 4 / PB steps; 256-wide groups fold their two halves
      // separately with the same scale): fold the exact integer partial sums into the fp32 accumulators
      if constexpr ((j + 1) % (4 / PB) == 0) {
        constexpr int ki = j / (4 / PB);
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) {
          float s, z = 0.f;
          if constexpr (FMT == 3) {
            s = 1.f;  // (the conversion has applied the scales)
          } else if constexpr (!is_int4) {
            const uint32_t e = sc[nt][ki];  // E8M0: 2^(byte - 127); byte 0 is the subnormal 2^-127
            s = __uint_as_float(e ? e << 23 : 0x00400000u);
          } else if constexpr (SV) {
            s = sv_get(sv_cur[nt], u);
            z = has_zp ? 16.0f + sv_get(zv_cur[nt], u) : 24.0f;
          } else {
            s = (float)sc[nt][ki];
            z = has_zp ? 16.0f + (float)zc[nt][ki] : 24.0f;
          }
          // (scalar FMAs; the file is built with -fno-slp-vectorize: left to itself the compiler forms v_pk_fma_f32 with
          // the scalar broadcast from the low half of a register PAIR whose high half it is free to use as the destination
          // of a ring load - the FMA then waits for a load it does not need, with the whole ring in flight behind it - and
          // packed f32 instructions issue slower next to MFMAs than the scalar ones they replace)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t = !is_int4 ? part[mt][nt][r] : __builtin_fmaf(-z, asum[mt][r], part[mt][nt][r]);
              acc[mt][nt][r] = __builtin_fmaf(s, t, acc[mt][nt][r]);
            }
            part[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asum[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
      }
      // the body is one basic block now: without fences the scheduler interleaves all steps and spills
      __builtin_amdgcn_sched_barrier(0);
    });
    }  // (probe 16)
    // Refill ring slot u with block kb + kD AFTER the last use of what it held: requested before (at the top of the
    // block) the new value had to live in other registers while the old one was still being expanded, and the loop end
    // moved all kD slots back into place with copies - each copy waits for the load into its source, so the whole ring was
    // drained once per trip of the loop (vmcnt(0) in front of 40 v_mov).
    load_w(kb + kD, wq_[u]);
    if constexpr (!SV) load_s(kb + kD, sq_[u], zq_[has_zp ? u : 0]);
    __builtin_amdgcn_sched_barrier(0);
  }
  }

  // ---- epilogue: lane owns out[m = 16 mt + 4 g + r][column of wave tile nt]
  if (gated) {
    if constexpr (NW >= 2) {
      constexpr int H2 = NW / 2;
#pragma unroll
      for (int nt = 0; nt < H2; ++nt) {
        const int n = out_col_of(nt);  // gate column = output column
        if (n >= Nh) continue;
        const float bg = bias ? bias[(int64_t)e * N + n] : 0.f, bu = bias ? bias[(int64_t)e * N + Nh + n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + 4 * g + r;
            float x = acc[mt][nt][r] + bg, y = acc[mt][nt + H2][r] + bu;
            if (fuse == 4) {
              x = fminf(x, act_limit);
              y = fminf(fmaxf(y, -act_limit), act_limit);
            }
            float a;
            if (fuse == 1 || fuse == 4) {
              a = x / (1.0f + expf(-x));
            } else {
              const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
              a = x * (0.5f * (1.0f + tanhf(inner)));
            }
            if (row < m_valid) out[(int64_t)(m0 + row) * Nh + n] = (T)(a * y);
          }
        }
      }
    }
    return;
  }
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
        if (pairs) {  // even lane: gate (weight row n), its right neighbour: up (row n + 1); N is even, so both or neither is in range
          const float other = __shfl_xor(v, 1, 64);
          if (!(l15 & 1) && row < m_valid) {
            const float gt = fminf(v, act_limit), up = fmaxf(-act_limit, fminf(other, act_limit)) + 1.0f;
            out[(int64_t)(m0 + row) * Nh + (n >> 1)] = (T)(gt * (1.0f / (1.0f + expf(-(gt * act_alpha)))) * up);
          }
          continue;
        }
        if (fuse == 3) {
          v = fmaxf(v, 0.f);
          v = v * v;
        }
        if (row < m_valid && !((probe & 2) && acc[mt][nt][r] != 12345.f)) out[(int64_t)(m0 + row) * N + n] = (T)v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Few workgroups, long K (the Mixtral down projection with one or two active experts: 64 column blocks for 256 CUs, 48 us
// for 59 MB): a workgroup owns 32 columns and its four waves split K into quarters. Every wave is on its own - its 16 x 128
// activation slices go through a wave-private LDS image (global -> registers four blocks ahead -> LDS one block ahead;
// LDS instructions of a wave execute in order, so no barrier), its weights through the same register ring as above - and the
// four partial sums meet in LDS once, added in a fixed order by wave 0. int4 codes, groups of 128, 16 rows, K % 2048 == 0.
// SPLIT = false: the same independent waves without the K split - a workgroup owns 128 columns, every wave 32 of them over
// all of K (no reduction): the barrier-free variant of the 16-row tile above.
template <typename T, int FMT, bool SPLIT = true>
__global__ __launch_bounds__(256, 2) void moe_w4a16_ksplit_kernel(T* __restrict__ out, const T* __restrict__ act,
                                                                  const uint8_t* __restrict__ wq, const void* __restrict__ scales_,
                                                                  const void* __restrict__ zeros_, const float* __restrict__ bias,
                                                                  const int32_t* __restrict__ rows_per_expert, int E, int N, int K) {
  constexpr int NW = 2, kD = 4, BN = SPLIT ? 16 * NW : 64 * NW, AROW = 256, kImg = 16 * AROW;
  constexpr bool has_zp = FMT == 2;
  constexpr bool kFp4 = FMT == 3;  // mxfp4 through the conversion instruction (bf16): scales = one dword of E8M0 bytes per block
  static_assert(!kFp4 || std::is_same<T, bf16>::value, "the fp4 conversion path is bf16 only");
  using S = typename std::conditional<kFp4, uint32_t, T>::type;
  __shared__ __attribute__((aligned(256))) char smem[4 * 2 * kImg];  // [wave][buffer]: 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const MoeTile tile = find_moe_tile(rows_per_expert, E, 16, (N + BN - 1) / BN);
  if (tile.expert < 0) return;
  const int e = tile.expert, m0 = tile.m0, m_valid = tile.m_valid;
  const int n_base = tile.col_block * BN + (SPLIT ? 0 : wave * (16 * NW));
  const int kgroups = K >> 7;
  const int per = SPLIT ? kgroups >> 2 : kgroups;  // 128-deep blocks of this wave (a multiple of kD)
  const int kb_off = SPLIT ? wave * per : 0;
  const S* scales = reinterpret_cast<const S*>(scales_);
  const S* zeros = reinterpret_cast<const S*>(zeros_);
  const uint8_t* wexp = wq + (int64_t)e * N * (K / 2);
  const S* sexp = scales + (int64_t)e * N * kgroups;
  const S* zexp = has_zp ? zeros + (int64_t)e * N * kgroups : sexp;
  uint32_t woff[NW], soff[NW];
#pragma unroll
  for (int nt = 0; nt < NW; ++nt) {
    int n = n_base + nt * 16 + l15;
    n = n < N ? n : N - 1;
    woff[nt] = (uint32_t)n * (uint32_t)(K / 2) + 16 * g + (uint32_t)kb_off * 64u;
    soff[nt] = (uint32_t)n * (uint32_t)kgroups + (uint32_t)kb_off;
  }
  // activation slice of a block: 16 rows x 16 chunks of 16 bytes, chunk q = i * 64 + lane
  const T* act_blk = act + (int64_t)m0 * K + (int64_t)kb_off * 128;
  char* img = smem + wave * (2 * kImg);
  uint32_t aoff[4], loff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 64 + lane, row = q >> 4, c = q & 15;
    aoff[i] = (uint32_t)(row < m_valid ? row : m_valid - 1) * (uint32_t)K + c * 8;
    const int slot = ((0x84C0 >> (c & 12)) & 15) + (c & 3);  // (the permuted-k slots of the kernel above)
    loff[i] = (uint32_t)(row * AROW + ((slot ^ (row & 15)) << 4));
  }
  auto load_a = [&](int kb, v4i (&r)[4]) {
    const uint32_t ko = (uint32_t)(kb < per ? kb : 0) * 128u;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const v4i*>(act_blk + (aoff[i] + ko));
  };
  auto store_a = [&](int buf, const v4i (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const v4i v = r[i];
      v4i p = v;  // int4: element order (a0,a4,a1,a5,a2,a6,a3,a7) to match expand_nibbles; fp4: the natural order
      if constexpr (!kFp4) {
        p[0] = (int)__builtin_amdgcn_perm((uint32_t)v[2], (uint32_t)v[0], 0x05040100u);
        p[1] = (int)__builtin_amdgcn_perm((uint32_t)v[2], (uint32_t)v[0], 0x07060302u);
        p[2] = (int)__builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[1], 0x05040100u);
        p[3] = (int)__builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[1], 0x07060302u);
      }
      *reinterpret_cast<v4i*>(img + buf * kImg + loff[i]) = p;
    }
  };
  uint32_t wq_[kD][NW][4];
  auto load_w = [&](int kb, uint32_t (&dst)[NW][4]) {
    const uint32_t koff = (uint32_t)(kb < per ? kb : 0) * 64u;
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      const v4i t = *reinterpret_cast<const v4i*>(wexp + woff[nt] + koff);
      dst[nt][0] = t[0]; dst[nt][1] = t[1]; dst[nt][2] = t[2]; dst[nt][3] = t[3];
    }
  };
  typedef uint32_t SVec __attribute__((ext_vector_type(kFp4 ? 4 : 2)));  // the kD = 4 scales of a trip: one 8- / 16-byte load
  SVec sv_cur[NW], sv_nxt[NW], zv_cur[NW], zv_nxt[NW];
  auto load_sv = [&](int kb0, SVec (&sd)[NW], SVec (&zd)[NW]) {
    const int k0 = kb0 < per ? kb0 : 0;
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      sd[nt] = *reinterpret_cast<const SVec*>(sexp + soff[nt] + k0);
      if constexpr (has_zp) zd[nt] = *reinterpret_cast<const SVec*>(zexp + soff[nt] + k0);
    }
  };
  auto sv_get = [&](const SVec& v, int u) -> float {
    const uint32_t d = v[u >> 1];
    const uint16_t h = (u & 1) ? (uint16_t)(d >> 16) : (uint16_t)d;
    return (float)__builtin_bit_cast(T, h);
  };
  v4f acc[NW], part[NW], asum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < NW; ++nt) {
    acc[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
    part[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
  }
  const v4i ones = {(int)W4<T>::kOnes, (int)W4<T>::kOnes, (int)W4<T>::kOnes, (int)W4<T>::kOnes};
  uint32_t magic = W4<T>::kMagic, nib_mask = (0xFu << W4<T>::kShift) | (0xFu << (W4<T>::kShift + 16));
  asm volatile("" : "+v"(magic), "+s"(nib_mask));

  v4i aq_[kD][4];
#pragma unroll
  for (int d = 0; d < kD; ++d) load_a(d, aq_[d]);
  store_a(0, aq_[0]);
  load_a(kD, aq_[0]);
  load_sv(0, sv_cur, zv_cur);
  load_sv(kD, sv_nxt, zv_nxt);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int d = 0; d < kD; ++d) {
    load_w(d, wq_[d]);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int kb0 = 0; kb0 < per; kb0 += kD) {
#pragma unroll
    for (int u = 0; u < kD; ++u) {
      const int kb = kb0 + u, buf = kb & 1;
      if (u == 0 && kb0 > 0) {
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) {
          sv_cur[nt] = sv_nxt[nt];
          if constexpr (has_zp) zv_cur[nt] = zv_nxt[nt];
        }
        load_sv(kb0 + kD, sv_nxt, zv_nxt);
      }
      // the image of block kb was written a block ago; this wave's own reads of the other buffer are done (in-order LDS)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      store_a(buf ^ 1, aq_[(u + 1) % kD]);
      load_a(kb + 1 + kD, aq_[(u + 1) % kD]);
      uint32_t wd[NW][4];
#pragma unroll
      for (int nt = 0; nt < NW; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) wd[nt][t] = FMT == 0 ? (wq_[u][nt][t] ^ 0x88888888u) : wq_[u][nt][t];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the stores above are in LDS before the reads below are issued)
      const char* abase = img + buf * kImg;
      static_for4([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        v4i wf[NW];
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) {
          if constexpr (kFp4) {
            const uint32_t e8 = ((uint32_t)sv_cur[nt][u] >> (8 * g)) & 0xffu;  // this lane's group: k = 128 kb + 32 g ..
            wf[nt] = expand_mxfp4_hw(wd[nt][j], __uint_as_float(e8 ? e8 << 23 : 0x00400000u));
          } else {
            wf[nt] = expand_nibbles<T>(wd[nt][j], nib_mask, magic);
          }
        }
        const int slot = ((0x84C0 >> (4 * g)) & 15) + j;
        const v4i af = *reinterpret_cast<const v4i*>(abase + l15 * AROW + ((slot ^ l15) << 4));
        if constexpr (!kFp4) asum = W4<T>::mma(af, ones, asum);
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) part[nt] = W4<T>::mma(af, wf[nt], part[nt]);
        if constexpr (j == 3) {
#pragma unroll
          for (int nt = 0; nt < NW; ++nt) {
            if constexpr (kFp4) {
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[nt][r] += part[nt][r];
            } else {
              const float sc = sv_get(sv_cur[nt], u);
              const float z = has_zp ? 16.0f + sv_get(zv_cur[nt], u) : 24.0f;
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[nt][r] = __builtin_fmaf(sc, __builtin_fmaf(-z, asum[r], part[nt][r]), acc[nt][r]);
            }
            part[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
          }
          asum = (v4f){0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      load_w(kb + kD, wq_[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the four partial sums meet in LDS (over the images: every wave is past its last read), fixed order
  if constexpr (SPLIT) {
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  if (wave != 0) {
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) *reinterpret_cast<v4f*>(&red[(((wave - 1) * NW + nt) * 64 + lane) * 4]) = acc[nt];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) {
      const v4f o = *reinterpret_cast<const v4f*>(&red[((w * NW + nt) * 64 + lane) * 4]);
      acc[nt][0] += o[0]; acc[nt][1] += o[1]; acc[nt][2] += o[2]; acc[nt][3] += o[3];
    }
  }
#pragma unroll
  for (int nt = 0; nt < NW; ++nt) {
    const int n = n_base + nt * 16 + l15;
    if (n >= N) continue;
    const float bv = bias ? bias[(int64_t)e * N + n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * g + r;
      if (row < m_valid) out[(int64_t)(m0 + row) * N + n] = (T)(acc[nt][r] + bv);
    }
  }
}

#ifdef SGLK_PROBES
static int g_w4_probe = 0, g_w4_mt = 0, g_w4_fp4hw = 1;
#else
constexpr int g_w4_probe = 0, g_w4_mt = 0, g_w4_fp4hw = 1;
#endif

// (the clamp bound of fused_act 4 rides beside the launch parameters: one value per call, set by the C-ABI entry)
static thread_local float t_act_limit = 0.f, t_act_alpha = 0.f;
static thread_local const int32_t* t_row_map = nullptr;  // the token gather of the call (streaming kernels only)
static thread_local int t_tail_flag = 0;  // kMoeTailFlag while the launches cover only the rows moe_persist.hip left over

template <typename T, int MT, int NW, int PB, int FMT = 0, int WV = 4>
static int launch_pb(hipStream_t st, void* out, const void* act, const void* wq, const void* scales, const void* zeros,
                  const float* bias, const int32_t* rows, int64_t total_m, int E, int N, int K, int group_shift, int fuse) {
  constexpr int BM = 16 * MT, BN = 16 * NW * WV;
  const bool gated = NW >= 2 && (fuse == 1 || fuse == 2 || fuse == 4);  // (5, the gpt-oss swiglu, runs on the plain layout)
  const int64_t wgs = moe_tile_launch_size(total_m, E, BM, gated ? cdiv(N / 2, BN / 2) : cdiv(N, BN));
  if (wgs >= ((int64_t)1 << 31)) return fail(SGLK_EINVAL, "moe_grouped_mm_nt_xe20_w4a16: problem too large for one launch");
  dim3 grid((unsigned)wgs);
  moe_w4a16_kernel<T, MT, NW, PB, FMT, WV><<<grid, 64 * WV, 0, st>>>((T*)out, (const T*)act, (const uint8_t*)wq, scales, zeros, bias,
                                                             rows, E | t_tail_flag, N, K, group_shift, g_w4_probe, fuse, t_act_limit, t_row_map, t_act_alpha);
  return check_launch("moe_grouped_mm_nt_xe20_w4a16");
}

template <typename T, int MT, int NW, int WV = 4>
static int launch(hipStream_t st, void* out, const void* act, const void* wq, const void* scales, const void* zeros,
                  const float* bias, const int32_t* rows, int64_t total_m, int E, int N, int K, int group_shift,
                  bool fp4hw = false, int fuse = 0) {
  if constexpr (std::is_same<T, bf16>::value) {
    if (fp4hw)  // mxfp4 through the conversion instruction: one dword of scale bytes per 128-deep block
      return launch_pb<T, MT, NW, 1, 3, WV>(st, out, act, wq, scales, nullptr, bias, rows, total_m, E, N, K, 7, fuse);
  }
  if (group_shift < 0)  // mxfp4: E8M0 scales per 32
    return launch_pb<T, MT, NW, 4, 1, WV>(st, out, act, wq, scales, nullptr, bias, rows, total_m, E, N, K, 5, fuse);
#define SGLK_W4_GO(PB)                                                                                                    \
  return zeros != nullptr ? launch_pb<T, MT, NW, PB, 2, WV>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fuse) \
                          : launch_pb<T, MT, NW, PB, 0, WV>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fuse)
  if (group_shift == 5) SGLK_W4_GO(4);
  if (group_shift == 6) SGLK_W4_GO(2);
  SGLK_W4_GO(1);
#undef SGLK_W4_GO
}

template <typename T>
static int dispatch(hipStream_t st, void* out, const void* act, const void* wq, const void* scales, const void* zeros,
                    const float* bias, const int32_t* rows, int64_t total_m, int E, int N, int K, int group_shift, int fuse) {
  // Tile policy by average rows per expert (the reference switches policies the same way,
  // GroupGemmW4A16Xe20.cpp:266-277). The row counts are ragged around the average, and a second row block of an expert
  // streams its weights again, so a tile is chosen that holds ~1.5x the average; 64-row tiles are the largest whose K loop
  // stays free of register spills.
  // mxfp4 with bf16 activations goes through the conversion instruction and the group-128 machinery (FMT 3)
  const bool fp4hw = group_shift < 0 && std::is_same<T, bf16>::value && K % 128 == 0 && (uintptr_t)scales % 4 == 0 && g_w4_fp4hw != 0;
  const int gp = fp4hw ? 7 : group_shift;  // the group shift the tile policy sees
  // (tail mode: total_m is the worst case that sizes the launch - 128 rows per expert; the tiles are chosen for what the tails
  // of routed experts look like, a few dozen rows)
  const int64_t pol_m = t_tail_flag ? std::min<int64_t>(total_m, 24 * (int64_t)E) : total_m;
  const int64_t avg = g_w4_mt ? ((g_w4_mt == 1 || g_w4_mt == 11 || g_w4_mt == 12 || g_w4_mt == 13) ? 1 : g_w4_mt == 2 ? 32 : g_w4_mt == 4 ? 200 : 1000) : pol_m / E;
  // (16-column tiles per wave - 64 columns per workgroup, twice the workgroups - were slower at every decode shape: the
  // activation staging and the barrier are per workgroup, 155 vs 145 us at N = 28672, K = 4096)
  // (the 16 / 32-row tiles fetch the scales of four 128-deep blocks with one 8-byte load: groups of 128, K % 512 == 0)
  const bool small_ok = gp != 7 && gp != 8 ? true : (gp == 7 && K % 512 == 0);
  if (!small_ok) return launch<T, 4, 2>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fp4hw, fuse);
  // measured on ragged counts around the average (gate/up + down projection of Mixtral, us): avg 8: 16-row tile 124 + 75,
  // 32-row 140 + 98; avg 16: 151 + 97 vs 145 + 99; avg 32: 226 + 130 vs 191 + 141 vs 64-row 251 + 175
  // few column blocks (the Mixtral down projection: 32 of 128 columns x 8 experts = one workgroup per CU): 64-column
  // workgroups with an 8-deep weight ring instead - twice the workgroups, the same bytes in flight per wave
  // (a narrow projection keeps the 16-row tile up to an average of 20 rows: ragged counts around 16 - 64 tokens, top-2 of 8 -
  // 81 us against 96 us with the 32-row tile for the down projection; the gate / up projection measured 151 against 145)
  const bool gated_epi = fuse == 1 || fuse == 2 || fuse == 4;  // (needs the two-tile wave: no 16-column wave tiles, no K split)
  const bool narrow16 = !gated_epi && gp == 7 && K % 1024 == 0 &&
                        std::max<int64_t>(std::min<int64_t>(pol_m, E), pol_m / 16) * cdiv(N, 128) <= 384;
  const bool small = avg <= 10 || (narrow16 && avg <= 20 && g_w4_mt == 0);
  const int64_t bm = small ? 16 : 32;
  const int64_t est_row_blocks = std::max<int64_t>(std::min<int64_t>(pol_m, E), pol_m / bm);
  const bool narrow = !gated_epi && gp == 7 && K % 1024 == 0 && (est_row_blocks * cdiv(N, 128) <= 384 || g_w4_mt == 11);
  if (small) {
    // fewer workgroups than CUs even with 64-column tiles, long K: four waves split K (see moe_w4a16_ksplit_kernel)
    // (not with a row_map: the K-split kernel reads expert-contiguous rows - a mapped call with fused_act = 0 computed on the
    //  wrong rows and read past the [src_rows, K] activations)
    const bool ksplit = t_row_map == nullptr && fuse == 0 && (group_shift == 7 || fp4hw) && K % 2048 == 0 && ((K >= 8192 && avg <= 10 && est_row_blocks * cdiv(N, 64) <= 1024) || g_w4_mt == 12) && g_w4_mt != 11 && g_w4_mt != 1 && g_w4_mt != 13;
    if (ksplit) {
      const int64_t wgs = moe_tile_launch_size(total_m, E, 16, cdiv(N, 32));
      if (wgs < ((int64_t)1 << 31)) {
        bool done = false;
        if constexpr (std::is_same<T, bf16>::value) {
          if (fp4hw) {
            moe_w4a16_ksplit_kernel<T, 3><<<dim3((unsigned)wgs), 256, 0, st>>>((T*)out, (const T*)act, (const uint8_t*)wq, scales, nullptr, bias, rows, E | t_tail_flag, N, K);
            done = true;
          }
        }
        if (done) {
        } else if (zeros != nullptr)
          moe_w4a16_ksplit_kernel<T, 2><<<dim3((unsigned)wgs), 256, 0, st>>>((T*)out, (const T*)act, (const uint8_t*)wq, scales, zeros, bias, rows, E | t_tail_flag, N, K);
        else
          moe_w4a16_ksplit_kernel<T, 0><<<dim3((unsigned)wgs), 256, 0, st>>>((T*)out, (const T*)act, (const uint8_t*)wq, scales, zeros, bias, rows, E | t_tail_flag, N, K);
        return check_launch("moe_grouped_mm_nt_xe20_w4a16");
      }
    }
#ifdef SGLK_PROBES  // (diagnostic build: the independent waves without the K split - gate / up projection 114-143 -> 119-158 us,
    // down projection at 16 rows per expert 72 -> 62 us against 67 us with the split)
    if (fuse == 0 && g_w4_mt == 13 && group_shift == 7 && K % 512 == 0) {
      const int64_t wgs = moe_tile_launch_size(total_m, E, 16, cdiv(N, 128));
      if (zeros != nullptr)
        moe_w4a16_ksplit_kernel<T, 2, false><<<dim3((unsigned)wgs), 256, 0, st>>>((T*)out, (const T*)act, (const uint8_t*)wq, scales, zeros, bias, rows, E | t_tail_flag, N, K);
      else
        moe_w4a16_ksplit_kernel<T, 0, false><<<dim3((unsigned)wgs), 256, 0, st>>>((T*)out, (const T*)act, (const uint8_t*)wq, scales, zeros, bias, rows, E | t_tail_flag, N, K);
      return check_launch("moe_grouped_mm_nt_xe20_w4a16");
    }
#endif
    if (narrow) return launch<T, 1, 1>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fp4hw, fuse);
    // (eight blocks in flight per wave at 128 columns: 175 registers, two waves per SIMD instead of three - no faster)
    return launch<T, 1, 2>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fp4hw, fuse);
  }
  // (128-row tiles and 64-column wave tiles both measured slower at 512 rows per expert: 1 wave per SIMD; the prefill side is
  // bound by L2 traffic - 64 x 128 tiles re-read activations 224 times and weights 8 times, 11 GB at ~10 TB/s)
  // eight waves share a staged activation tile (64 x 256): half the activation traffic and barriers per flop of the 64 x 128
  // tile - 512 rows per expert 1.90 -> 1.61 ms (gate / up), 0.84 -> 0.73 ms (down); 128 rows 0.49 (0.48 with 32-row tiles) -> 0.43 ms
  if (g_w4_mt == 8 || (g_w4_mt == 0 && avg >= 112 && N % 256 == 0))
    return launch<T, 4, 2, 8>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fp4hw, fuse);
  if (avg <= 160) {  // (avg 64: 32-row tile 321 + 198 us, 64-row 352 + 230; avg 128: 571 + 352 vs 596 + 396; 256: 1021 + 615 vs 1008 + 471)
    if (narrow) return launch<T, 2, 1>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fp4hw, fuse);
    return launch<T, 2, 2>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fp4hw, fuse);
  }
  return launch<T, 4, 2>(st, out, act, wq, scales, zeros, bias, rows, total_m, E, N, K, group_shift, fp4hw, fuse);
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_moe_grouped_mm_w4a16(sglk_stream_t stream, void* out, const void* activations,
                                         const void* packed_weights, const void* scales, const void* zeros,
                                         const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                         int64_t n_experts, int64_t N, int64_t K, int64_t group_size, int is_int4,
                                         int dtype) {
  return sglk_moe_grouped_mm_w4a16_act(stream, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, total_m,
                                       n_experts, N, K, group_size, is_int4, dtype, 0, 0.f, nullptr, 0);
}

static int w4a16_run(sglk_stream_t stream, void* out, const void* activations, const void* packed_weights, const void* scales,
                     const void* zeros, const float* bias, const int32_t* rows_per_expert, int64_t total_m, int64_t n_experts,
                     int64_t N, int64_t K, int64_t group_size, int is_int4, int dtype, int fused_act, float act_limit,
                     const int32_t* row_map, int64_t src_rows, float* split_ws, int* split_used, float act_alpha = 0.f);

extern "C" int sglk_moe_grouped_mm_w4a16_act(sglk_stream_t stream, void* out, const void* activations,
                                             const void* packed_weights, const void* scales, const void* zeros,
                                             const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                             int64_t n_experts, int64_t N, int64_t K, int64_t group_size, int is_int4,
                                             int dtype, int fused_act, float act_limit, const int32_t* row_map,
                                             int64_t src_rows) {
  // (5, the gpt-oss swiglu, needs its alpha: sglk_moe_grouped_mm_w4a16_swiglu)
  SGLK_REQUIRE(fused_act >= 0 && fused_act <= 4,
               "moe_grouped_mm_nt_xe20_w4a16: fused_act must be 0 (none), 1 (silu), 2 (gelu), 3 (relu2) or 4 (clamped swiglu)");
  return w4a16_run(stream, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, total_m, n_experts, N, K,
                   group_size, is_int4, dtype, fused_act, act_limit, row_map, src_rows, nullptr, nullptr);
}

extern "C" int sglk_moe_grouped_mm_w4a16_swiglu(sglk_stream_t stream, void* out, const void* activations,
                                                const void* packed_weights, const void* scales, const void* zeros,
                                                const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                                int64_t n_experts, int64_t N, int64_t K, int64_t group_size, int is_int4,
                                                int dtype, float alpha, float limit, const int32_t* row_map, int64_t src_rows) {
  return w4a16_run(stream, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, total_m, n_experts, N, K,
                   group_size, is_int4, dtype, 5, limit, row_map, src_rows, nullptr, nullptr, alpha);
}

extern "C" int sglk_moe_grouped_mm_w4a16_splitk(sglk_stream_t stream, void* out, float* ws, const void* activations,
                                                const void* packed_weights, const void* scales, const void* zeros,
                                                const int32_t* rows_per_expert, int64_t total_m, int64_t n_experts,
                                                int64_t N, int64_t K, int64_t group_size, int is_int4, int dtype,
                                                int* split_used) {
  SGLK_REQUIRE(split_used != nullptr, "moe_grouped_mm_nt_w4a16_splitk: split_used must not be NULL");
  *split_used = 0;
  return w4a16_run(stream, out, activations, packed_weights, scales, zeros, nullptr, rows_per_expert, total_m, n_experts, N, K,
                   group_size, is_int4, dtype, 0, 0.f, nullptr, 0, ws, split_used);
}

extern "C" int sglk_moe_w4a16_splitk_applies(int64_t total_m, int64_t n_experts, int64_t N, int64_t K, int64_t group_size,
                                             int is_int4, int dtype) {
  if (total_m <= 0 || n_experts <= 0 || N <= 0 || K <= 0 || N >= (1ll << 31) || K >= (1ll << 31) || n_experts >= (1ll << 20)) return 0;
  const int gs = !is_int4 ? 5 : group_size == 32 ? 5 : group_size == 64 ? 6 : group_size == 128 ? 7 : group_size == 256 ? 8 : -1;
  if (gs < 0) return 0;
  return sglk::moe_persist_splitk_applies(total_m, (int)n_experts, (int)N, (int)K, gs, is_int4 ? 1 : 2, dtype);
}

static int w4a16_run(sglk_stream_t stream, void* out, const void* activations, const void* packed_weights, const void* scales,
                     const void* zeros, const float* bias, const int32_t* rows_per_expert, int64_t total_m, int64_t n_experts,
                     int64_t N, int64_t K, int64_t group_size, int is_int4, int dtype, int fused_act, float act_limit,
                     const int32_t* row_map, int64_t src_rows, float* split_ws, int* split_used, float act_alpha) {
  using namespace sglk;
  SGLK_REQUIRE(row_map == nullptr || (src_rows > 0 && src_rows * K < (1ll << 32)),
               "moe_grouped_mm_nt_xe20_w4a16: a row map needs 0 < src_rows and src_rows * K < 2^32 (src_rows=%lld)",
               (long long)src_rows);
  SGLK_REQUIRE(fused_act >= 0 && fused_act <= 5,
               "moe_grouped_mm_nt_xe20_w4a16: fused_act must be 0 (none), 1 (silu), 2 (gelu), 3 (relu2), 4 (clamped swiglu) or 5 "
               "(gpt-oss swiglu)");
  SGLK_REQUIRE((fused_act != 4 && fused_act != 5) || act_limit > 0.f,
               "moe_grouped_mm_nt_xe20_w4a16: the clamped / gpt-oss swiglu needs a positive limit");
  t_act_limit = act_limit;
  t_act_alpha = act_alpha;
  SGLK_REQUIRE(!(fused_act == 1 || fused_act == 2 || fused_act == 4 || fused_act == 5) || N % 16 == 0,
               "moe_grouped_mm_nt_xe20_w4a16: a gated epilogue needs N (gate + up rows) to be a multiple of 16");
  SGLK_REQUIRE(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256,
               "group_size must be 32, 64, 128 or 256; got %lld", (long long)group_size);
  SGLK_REQUIRE(is_int4 || (group_size == 32 && zeros == nullptr),
               "moe_grouped_mm_nt_xe20_w4a16: mxfp4 weights use E8M0 scales per 32 elements and no zero points");
  SGLK_REQUIRE(K > 0 && K % group_size == 0, "K must be a multiple of group_size");
  SGLK_REQUIRE(N > 0 && N % 8 == 0, "N must be divisible by 8");
  SGLK_REQUIRE(n_experts > 0, "n_experts must be positive");
  SGLK_REQUIRE(dtype == SGLK_BF16 || dtype == SGLK_F16, "activations must be bfloat16 or half");
  SGLK_REQUIRE((uintptr_t)activations % 16 == 0 && (uintptr_t)packed_weights % 16 == 0,
               "moe_grouped_mm_nt_xe20_w4a16: activations and packed_weights must be 16-byte aligned");
  // (the decode tiles fetch a row's scales / zero points of a whole trip of the K loop with one 8- to 32-byte load)
  SGLK_REQUIRE((uintptr_t)scales % 16 == 0 && (zeros == nullptr || (uintptr_t)zeros % 16 == 0),
               "moe_grouped_mm_nt_xe20_w4a16: scales and zeros must be 16-byte aligned");
  // (per-lane weight / activation / scale offsets inside one expert are 32-bit)
  SGLK_REQUIRE(N * (K / 2) < (1ll << 32) && 256 * K < (1ll << 32) && N < (1ll << 31) && K < (1ll << 31),
               "moe_grouped_mm_nt_xe20_w4a16: one expert's packed weights must stay below 4 GiB (N=%lld, K=%lld)",
               (long long)N, (long long)K);
  if (total_m == 0) return SGLK_OK;
  const int gs = !is_int4 ? -1 : group_size == 32 ? 5 : group_size == 64 ? 6 : group_size == 128 ? 7 : 8;
  hipStream_t st = (hipStream_t)stream;
  // (a row map is honoured by the streaming kernels' staging loads; the tile pipeline stages by LDS-DMA from expert-contiguous
  // rows, so a mapped call stays on the streaming kernels at every size - fused_experts maps only below 96 rows per expert)
  struct MapScope {
    explicit MapScope(const int32_t* m) { t_row_map = m; }
    ~MapScope() { t_row_map = nullptr; }
  } map_scope(row_map);
  if (row_map == nullptr && split_ws != nullptr && bias == nullptr && fused_act == 0) {
    // the K split of the down projection: the full 128-row blocks as fp32 partial sums in split_ws, remainders of 1 .. 64 rows
    // on the streaming kernels into `out`
    int rc = moe_persist_splitk_try(st, split_ws, activations, packed_weights, scales, is_int4 ? zeros : nullptr, gs, rows_per_expert,
                                    total_m, (int)n_experts, (int)N, (int)K, dtype, is_int4 ? 1 : 2);
    if (rc < 0) return rc;
    if (rc > 0) {
      *split_used = rc == 2 ? 128 : 256;
      t_tail_flag = rc == 2 ? kMoeTailFlag128 : kMoeTailFlag;
      const int64_t tail_m = std::min<int64_t>(total_m, (rc == 2 ? 64 : 128) * n_experts);
      rc = dtype == SGLK_BF16 ? dispatch<bf16>(st, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, tail_m,
                                               (int)n_experts, (int)N, (int)K, gs, fused_act)
                              : dispatch<f16>(st, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, tail_m,
                                              (int)n_experts, (int)N, (int)K, gs, fused_act);
      t_tail_flag = 0;
      return rc;
    }
  }
  if (row_map == nullptr) {
    if (int rc = moe_persist_try(st, out, activations, packed_weights, scales, is_int4 ? zeros : nullptr, gs, bias, rows_per_expert, total_m, (int)n_experts,
                                 (int)N, (int)K, 0, 0, dtype, is_int4 ? 1 : 2, fused_act, act_limit, act_alpha)) {
      if (rc < 0) return rc;
      if (rc == 3) return SGLK_OK;  // (the tile pipeline took the remainders too)
      // the experts' last rows (at most 128 each) on the streaming kernels, sized for the worst case
      t_tail_flag = rc == 2 ? kMoeTailFlag128 : kMoeTailFlag;
      const int64_t tail_m = std::min<int64_t>(total_m, (rc == 2 ? 64 : 128) * n_experts);
      rc = dtype == SGLK_BF16 ? dispatch<bf16>(st, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, tail_m,
                                               (int)n_experts, (int)N, (int)K, gs, fused_act)
                              : dispatch<f16>(st, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, tail_m,
                                              (int)n_experts, (int)N, (int)K, gs, fused_act);
      t_tail_flag = 0;
      return rc;
    }
  }
  if (dtype == SGLK_BF16)
    return dispatch<bf16>(st, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, total_m,
                          (int)n_experts, (int)N, (int)K, gs, fused_act);
  return dispatch<f16>(st, out, activations, packed_weights, scales, zeros, bias, rows_per_expert, total_m,
                       (int)n_experts, (int)N, (int)K, gs, fused_act);
}

#ifdef SGLK_PROBES
extern "C" SGLK_API void sglk_debug_set_w4a16_probe(int probe, int force_mt) {
  sglk::g_w4_probe = probe;
  sglk::g_w4_mt = force_mt;
}
#endif
