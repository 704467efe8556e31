// qserve_w4a8_per_chn_gemm / qserve_w4a8_per_group_gemm for gfx950 (QServe W4A8: int8 activations, uint4 weights).
//
// The reference only declares these ops (include/sgl_kernel_ops.h:1132-1148; wrappers python/sgl_kernel/gemm.py:
// 314-356); their meaning and the weight / scale packing are pinned by tests/test_qserve_w4a8_per_chn_gemm.py:12-56,
// 80-88 and tests/test_qserve_w4a8_per_group_gemm.py:12-92,134-145:
//   per channel: out = fp16( (Aq @ Wq^T) * sA[m] * sW[n] - a_ssum[m] * w_szs[n] ),  w_szs = zero * sW
//   per group  : out = fp16( (Aq @ W8^T) * sA[m] * sW[n] ),  W8[n,k] = Wq[n,k] * s8[k/128, n] + zs8[k/128, n]
//                (int8 two-level "progressive" quantisation: s8 int8 group scale, zs8 = -zero * s8, |W8| <= 127)
// Packing (convert_to_qserve_format): Wq [N,K] uint4 is cut in 32 x 32 blocks of 512 bytes, block (n/32, k/32) at
// byte (n/32 * K/32 + k/32) * 512; inside a block byte ((c*4 + e)*2 + d)*2*4 + b*4 + f holds k = 16 d + 4 e + f of
// rows 8 b + c (low nibble) and 16 + 8 b + c (high nibble). s8 / zs8 rows are permuted per 32 columns: position
// 4 c + q holds column 8 q + c.
//
// Kernel: that layout IS an MFMA layout. v_mfma_i32_16x16x64_i8 wants from lane (j, kg) 16 consecutive k of row j:
// with j = 8 b + c and kg = (k/32 odd, d) the four dwords e = 0..3 of the lane (16 bytes apart) hold k = 16 kg ..
// 16 kg + 15 of row j in their low nibbles and of row 16 + j in their high nibbles. Two AND / shift per dword unpack
// two A operands; the activations are the B operand (lane (m, kg): 16 bytes of row m straight from global memory,
// the four waves of a workgroup hit the same lines in the vector L1), so a lane ends up with 4 consecutive n of
// one m: 8-byte fp16 stores. The integer dot products are exact in int32; per group the weights are rebuilt as
// int8 in registers (packed multiply by the small group scale, carry-free packed add of zs8) before the MFMA, so
// both variants accumulate over all of K without rescaling.
// (That lane map is what the tile kernels below use; the decode kernel reads the same bytes another way, see there.)
#include <algorithm>

#include "common.h"

namespace sglk {

// gemm_8bit.hip: the persistent 256 x 256 tile pipeline with the packed weights staged in LDS (many rows)
int qserve_w4a8_persist(hipStream_t st, bool group, void* out, const void* a, const void* w, const void* zeros,
                        const void* scales_i8, const void* wscales, const void* ascales, const void* w_szs,
                        const void* a_ssums, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc);

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

// byte-wise a + b (mod 256) on four packed bytes
__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t b) {
  return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}

// Few rows (M <= 64: the weight stream of a decode step). Round 3's kernel gave a lane the four dwords e = 0..3 of "its" row
// as the packing suggests: 4-byte loads 16 bytes apart (every instruction touched all the lines of its 1 KiB and used a
// quarter of each), single-byte loads for the group scales, 16 waves of four short steps per workgroup - 24.5 us for the
// 29 MB of a 14336 x 4096 layer (0.15 of HBM), the waves waiting on a counter 0.5 - 0.6 of their cycles. Here a lane takes
// 32 CONTIGUOUS bytes of a block - chunks e = 2 p, 2 p + 1 of byte row c, i.e. k = {8 p .. 8 p + 7} and {16 + 8 p ..} of the four
// rows c, 8 + c (low nibbles, dwords b = 0 / 1) and 16 + c, 24 + c (high nibbles) - as two 16-byte loads, and supplies one
// of those four rows to each of FOUR MFMAs:
//   lane = (c = lane % 8, q0 = lane / 8 % 2, p = lane / 16 % 2, q1 = lane / 32): MFMA row j = 8 q0 + c, k group kg = 2 q1 + p;
//   q0 = which of the wave's two neighbouring 32-column blocks, q1 = which of the step's two 32-deep k blocks;
//   MFMA t (0..3) multiplies weight rows 8 t + c of both blocks: 16 rows x 64 k, the wave 64 columns x 64 k per step;
//   its A operand = dwords {X[b], Y[b], X[2 + b], Y[2 + b]} (X, Y = the two chunks, b = t % 2), nibble t / 2, which is k order
//   {8 p .. 8 p + 7, 16 + 8 p .. 16 + 8 p + 7} of k block 2 ks + q1: the B operand is two 8-byte loads of the activation row.
// The four group scales / zero terms a lane needs (columns c, 8 + c, 16 + c, 24 + c) are the four bytes of ONE dword of the
// permuted scale row (position 4 c + q holds column 8 q + c). A workgroup = KS waves on one 64-column pair, wave w takes the
// 64-deep steps w, w + KS, .. through a KD-deep static register ring (16 KiB of weights in flight per wave at KD = 8); the
// int32 partial sums meet in LDS and are added in wave order (exact). KS is chosen by the host for ~1500 waves per launch,
// KD so that the ring is no deeper than a wave's steps (slots past them would re-load the last step).
// (Measured and dropped: a wave-step of the four consecutive blocks of ONE 32-column group - 2 KiB contiguous, 128 deep, twice
// the workgroups - needs two MFMAs per row set, each with half of the row slots zeroed, and four activation loads per step:
// 11.3 against 7.8 us at N = 14336, K = 4096, one row.)
// Round 5 (17 - 64 rows were at 0.14 of HBM, 26 us for the 29 MB of a 14336 x 4096 layer):
//   * the B operand as ONE 16-byte load per lane and m-tile instead of two 8-byte loads: lane p = 0 takes bytes 0..15 of the
//     row's 32-byte k block, lane p = 1 (16 lanes on) bytes 16..31, and two v_permlane16_swap exchange the halves so that
//     p = 0 holds {0..7, 16..23} and p = 1 {8..15, 24..31}, the k order of the A operand (26.1 -> 17.0 us at 64 rows);
//   * separate ring depths: KD steps of weights (a wave's whole k range when it has at most eight steps: the weight stream
//     is one round trip), KA steps of activations (L2-resident, 16 registers per step and m-tile group);
//   * the K-split reduction and the stores are spread over all waves (each takes the output tiles t = wave, wave + KS, ..
//     and adds the KS int32 partial sums, 16-byte LDS accesses) instead of wave 0 doing 448 ds_read_b32 and every store.
template <bool GROUP, int MF, int KS, int KD, int KA, bool IL = true>
__global__ __launch_bounds__(64 * KS) void qserve_w4a8_stream_kernel(
    f16* __restrict__ out, const int8_t* __restrict__ a, const uint8_t* __restrict__ w,
    const int8_t* __restrict__ zeros, const int8_t* __restrict__ scales_i8, const f16* __restrict__ wscales,
    const f16* __restrict__ ascales, const f16* __restrict__ w_szs, const f16* __restrict__ a_ssums, int M, int N,
    int K, int64_t lda, int64_t ldc) {
  static_assert(KD % KA == 0, "the activation ring divides the weight ring");
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.x;
  const int m0 = blockIdx.y * (16 * MF);
  const int j = lane & 15, kg = lane >> 4;
  const int c = j & 7, q0 = j >> 3, p = kg & 1, q1 = kg >> 1;
  const int n32 = pair * 2 + q0;
  const int n32c = n32 * 32 < N ? n32 : pair * 2;  // (N = 32 mod 64: the pair's second block re-reads the first, not stored)

  const uint8_t* wl = w + ((int64_t)n32c * (K >> 5) + q1) * 512 + c * 64 + p * 32;
  const int8_t* al[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    int m = m0 + mf * 16 + j;
    m = m < M ? m : M - 1;
    al[mf] = a + (int64_t)m * lda + q1 * 32 + p * 16;
  }
  const int8_t* s8 = GROUP ? scales_i8 + n32c * 32 + c * 4 : nullptr;
  const int8_t* z8 = GROUP ? zeros + n32c * 32 + c * 4 : nullptr;

  v4i acc[MF][4];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mf][t] = (v4i){0, 0, 0, 0};

  const int nks_all = K >> 6;
  const int nks = nks_all > wave ? (nks_all - wave + KS - 1) / KS : 0;  // this wave's steps ks = wave + KS i
  v4i wq_[KD][2];
  v4i aq_[KA][MF];
  uint32_t sq_[KD][2];
  auto step_of = [&](int i) {
    i = i < nks ? i : nks - 1;
    int ks = wave + KS * (i > 0 ? i : 0);
    return ks < nks_all ? ks : nks_all - 1;  // (a wave without steps loads the last one; nothing is accumulated)
  };
  auto load_w = [&](int i, v4i (&wd)[2], uint32_t (&sz)[2]) {
    const int ks = step_of(i);
    wd[0] = *reinterpret_cast<const v4i*>(wl + (int64_t)ks * 1024);
    wd[1] = *reinterpret_cast<const v4i*>(wl + (int64_t)ks * 1024 + 16);
    if constexpr (GROUP) {
      const int64_t g = ks >> 1;
      sz[0] = *reinterpret_cast<const uint32_t*>(s8 + g * N);
      sz[1] = *reinterpret_cast<const uint32_t*>(z8 + g * N);
    }
  };
  auto load_a = [&](int i, v4i (&af)[MF]) {
    const int ks = step_of(i);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) af[mf] = *reinterpret_cast<const v4i*>(al[mf] + ks * 64);
  };
  // (a wave's loads return in issue order: the first steps' activations go out WITH their weights, not behind the whole ring)
#pragma unroll
  for (int d = 0; d < KD; ++d) {
    load_w(d, wq_[d], sq_[d]);
    if (IL && d < KA) load_a(d, aq_[d]);
  }
  if constexpr (!IL) {  // (probe: all weights first)
#pragma unroll
    for (int d = 0; d < KA; ++d) load_a(d, aq_[d]);
  }
  for (int ks0 = 0; ks0 < nks; ks0 += KD) {
#pragma unroll
    for (int u = 0; u < KD; ++u) {
      // (steps past this wave's range multiply zero weights: no branch; a slot is refilled AFTER its last use, else the new
      // step lives in other registers and the loop end moves the ring back with copies that wait for every load)
      const uint32_t keep = ks0 + u < nks ? 0xffffffffu : 0u;
      uint32_t lo[2][4], hi[2][4];  // [chunk][dword]: low / high nibbles as bytes
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t wd = (uint32_t)wq_[u][x][e] & keep;
          lo[x][e] = wd & 0x0f0f0f0fu;
          hi[x][e] = (wd >> 4) & 0x0f0f0f0fu;
        }
      v4i wop[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int b = t & 1;
        uint32_t d4[4] = {t < 2 ? lo[0][b] : hi[0][b], t < 2 ? lo[1][b] : hi[1][b], t < 2 ? lo[0][2 + b] : hi[0][2 + b],
                          t < 2 ? lo[1][2 + b] : hi[1][2 + b]};
        if constexpr (GROUP) {
          // code * s8 + (zs8 + 128) is in 1..255 for every byte (|code * s8 + zs8| <= 127): one packed 16-bit multiply-add per
          // dword, no carries between bytes; the XOR turns u8 + 128 back into int8
          const uint32_t sv = (sq_[u][0] >> (8 * t)) & 0xffu;
          const uint32_t zv = ((sq_[u][1] >> (8 * t)) & 0xffu) ^ 0x80u;
          const uint32_t smul = sv | (sv << 16);
          const uint32_t z2 = zv | (zv << 8);
          const uint32_t zadd = (z2 | (z2 << 16)) & keep;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            d4[e] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, d4[e]) * __builtin_bit_cast(u16x2, smul) +
                                                         __builtin_bit_cast(u16x2, zadd))) ^ (0x80808080u & keep);
        }
        wop[t] = (v4i){(int)d4[0], (int)d4[1], (int)d4[2], (int)d4[3]};
      }
      const int ua = u % KA;  // (u is an unrolled loop counter: a constant after unrolling)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        // lanes p = 0 / 1 of a row exchange halves: v_permlane16_swap swaps the odd 16-lane rows of its first operand (p = 1's
        // bytes 16..23) with the even rows of its second (p = 0's bytes 8..15)
        v4i bop = aq_[ua][mf];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3"
                     : "+v"(bop[0]), "+v"(bop[1]), "+v"(bop[2]), "+v"(bop[3]));
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mf][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wop[t], bop, acc[mf][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_w(ks0 + u + KD, wq_[u], sq_[u]);
      load_a(ks0 + u + KA, aq_[ua]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- K-split reduction (int32: exact in any order) and epilogue, spread over the waves: output tile ti = 4 mf + t goes to
  // wave ti % KS. Lane (column m = j of the m fragment; MFMA rows 4 kg + r = block kg / 2, byte row 4 (kg % 2) + r): weight row
  // t of that block's 8 t + c, i.e. columns n = 32 (2 pair + kg / 2) + 8 t + 4 (kg % 2) + r, r = 0..3 consecutive.
  extern __shared__ __attribute__((aligned(16))) int red[];  // [KS][MF * 4][64 lanes] x v4i
  v4i* red4 = reinterpret_cast<v4i*>(red);
  if constexpr (KS > 1) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if ((mf * 4 + t) % KS != wave) red4[((wave * MF + mf) * 4 + t) * 64 + lane] = acc[mf][t];  // (the owner keeps its own)
    __syncthreads();
  }
  const int nblk = pair * 2 + (kg >> 1);
  if (nblk * 32 >= N) return;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if ((mf * 4 + t) % KS != wave) continue;
      v4i sum = acc[mf][t];
      if constexpr (KS > 1) {
#pragma unroll
        for (int w2 = 0; w2 < KS; ++w2)
          if (w2 != wave) {
            const v4i o = red4[((w2 * MF + mf) * 4 + t) * 64 + lane];
            sum[0] += o[0]; sum[1] += o[1]; sum[2] += o[2]; sum[3] += o[3];
          }
      }
      const int m = m0 + mf * 16 + j;
      if (m >= M) continue;
      const float sa = (float)ascales[m];
      const float asum = GROUP ? 0.f : (float)a_ssums[m];
      const int n = nblk * 32 + t * 8 + (kg & 1) * 4;
      Vec<f16, 4> o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (float)sum[r] * sa * (float)wscales[n + r];
        if constexpr (!GROUP) v -= asum * (float)w_szs[n + r];
        o[r] = (f16)v;
      }
      store_vec<f16, 4>(out + (int64_t)m * ldc + n, o);
    }
  }
}

// 17 - 64 rows, wide N (round 5, second form): the stream above was bound by what its workgroups pull through L2 -> L1, not by HBM:
// with four 16-row m-tiles a wave fetches 4 KiB of activations per 2 KiB of weights, every one of the 224 workgroups the whole
// [64 x K] activation matrix - 57 MB beside the 29 MB of weights of a 14336 x 4096 layer. Here the SAME 32 contiguous bytes per
// lane feed v_mfma_i32_32x32x32_i8 instead: a wave covers FOUR neighbouring 32-column blocks (128 columns) x one 32-deep k block
// per step, the m-tiles are 32 rows - 2 KiB of activations per 2 KiB of weights, and half as many workgroups (each reads the
// activation matrix once: 29 MB).
//   lane = (c = lane % 8, r2 = lane / 8 % 4, h = lane / 32): MFMA row i = 8 r2 + c, k half h; the lane loads bytes [32 h, 32 h + 32)
//   of byte row c of block (4 quad + r2, k block ks): chunks e = 2 h, 2 h + 1, i.e. k = {8 h .. 8 h + 7} and {16 + 8 h ..} of the
//   rows c, 8 + c (low nibbles) and 16 + c, 24 + c (high nibbles); MFMA t multiplies weight rows 8 t + c of the four blocks with
//   A = {X[b], Y[b], X[2 + b], Y[2 + b]} (b = t % 2, nibble t / 2) as above; B = 16 bytes [16 h, 16 h + 16) of activation row
//   32 mf + lane % 32 with the halves exchanged between lanes l and l + 32 (v_permlane32_swap): {8 h .., 16 + 8 h ..}.
//   Accumulator register r of tile (mf, t): MFMA row 8 (r / 4) + 4 h + r % 4 = block r / 4, byte row 4 h + r % 4, i.e. output
//   columns 32 (4 quad + r / 4) + 8 t + 4 h + (0..3): four consecutive fp16 per store.
// A workgroup = 8 waves on one 128-column quad, wave w takes the 32-deep steps w, w + 8, ..; the int32 partial sums meet in LDS
// in two passes of 128 KiB (tiles 0..3, then 4..7: tile ti belongs to wave ti).
typedef int v16i __attribute__((ext_vector_type(16)));
template <bool GROUP, int MF2, int KD, int KA, int KS = 8>
__global__ __launch_bounds__(64 * KS) void qserve_w4a8_stream32_kernel(
    f16* __restrict__ out, const int8_t* __restrict__ a, const uint8_t* __restrict__ w,
    const int8_t* __restrict__ zeros, const int8_t* __restrict__ scales_i8, const f16* __restrict__ wscales,
    const f16* __restrict__ ascales, const f16* __restrict__ w_szs, const f16* __restrict__ a_ssums, int M, int N,
    int K, int64_t lda, int64_t ldc) {
  static_assert(KD % KA == 0, "the activation ring divides the weight ring");
  static_assert(KS == 8 || KS == 16, "8 or 16 waves split K");
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int quad = blockIdx.x;
  const int m0 = blockIdx.y * (32 * MF2);
  const int c = lane & 7, r2 = (lane >> 3) & 3, h = lane >> 5, i32 = lane & 31;
  const int n32 = quad * 4 + r2;
  const int n32c = n32 * 32 < N ? n32 : quad * 4;  // (a block past N re-reads the quad's first one and is not stored)

  const uint8_t* wl = w + (int64_t)n32c * (K >> 5) * 512 + c * 64 + h * 32;
  const int8_t* al[MF2];
#pragma unroll
  for (int mf = 0; mf < MF2; ++mf) {
    int m = m0 + mf * 32 + i32;
    m = m < M ? m : M - 1;
    al[mf] = a + (int64_t)m * lda + h * 16;
  }
  const int8_t* s8 = GROUP ? scales_i8 + n32c * 32 + c * 4 : nullptr;
  const int8_t* z8 = GROUP ? zeros + n32c * 32 + c * 4 : nullptr;

  v16i acc[MF2][4];
#pragma unroll
  for (int mf = 0; mf < MF2; ++mf)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][t][r] = 0;

  const int nks_all = K >> 5;                                            // 32-deep steps
  const int nks = nks_all > wave ? (nks_all - wave + KS - 1) / KS : 0;   // this wave's steps ks = wave + KS i
  v4i wq_[KD][2];
  v4i aq_[KA][MF2];
  uint32_t sq_[KD][2];
  auto step_of = [&](int i) {
    i = i < nks ? i : nks - 1;
    int ks = wave + KS * (i > 0 ? i : 0);
    return ks < nks_all ? ks : nks_all - 1;
  };
  auto load_w = [&](int i, v4i (&wd)[2], uint32_t (&sz)[2]) {
    const int ks = step_of(i);
    wd[0] = *reinterpret_cast<const v4i*>(wl + (int64_t)ks * 512);
    wd[1] = *reinterpret_cast<const v4i*>(wl + (int64_t)ks * 512 + 16);
    if constexpr (GROUP) {
      const int64_t g = ks >> 2;
      sz[0] = *reinterpret_cast<const uint32_t*>(s8 + g * N);
      sz[1] = *reinterpret_cast<const uint32_t*>(z8 + g * N);
    }
  };
  auto load_a = [&](int i, v4i (&af)[MF2]) {
    const int ks = step_of(i);
#pragma unroll
    for (int mf = 0; mf < MF2; ++mf) af[mf] = *reinterpret_cast<const v4i*>(al[mf] + ks * 32);
  };
#pragma unroll
  for (int d = 0; d < KD; ++d) {
    load_w(d, wq_[d], sq_[d]);
    if (d < KA) load_a(d, aq_[d]);
  }
  for (int ks0 = 0; ks0 < nks; ks0 += KD) {
#pragma unroll
    for (int u = 0; u < KD; ++u) {
      const uint32_t keep = ks0 + u < nks ? 0xffffffffu : 0u;
      uint32_t lo[2][4], hi[2][4];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t wd = (uint32_t)wq_[u][x][e] & keep;
          lo[x][e] = wd & 0x0f0f0f0fu;
          hi[x][e] = (wd >> 4) & 0x0f0f0f0fu;
        }
      v4i wop[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int b = t & 1;
        uint32_t d4[4] = {t < 2 ? lo[0][b] : hi[0][b], t < 2 ? lo[1][b] : hi[1][b], t < 2 ? lo[0][2 + b] : hi[0][2 + b],
                          t < 2 ? lo[1][2 + b] : hi[1][2 + b]};
        if constexpr (GROUP) {
          const uint32_t sv = (sq_[u][0] >> (8 * t)) & 0xffu;
          const uint32_t zv = ((sq_[u][1] >> (8 * t)) & 0xffu) ^ 0x80u;
          const uint32_t smul = sv | (sv << 16);
          const uint32_t z2 = zv | (zv << 8);
          const uint32_t zadd = (z2 | (z2 << 16)) & keep;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            d4[e] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, d4[e]) * __builtin_bit_cast(u16x2, smul) +
                                                         __builtin_bit_cast(u16x2, zadd))) ^ (0x80808080u & keep);
        }
        wop[t] = (v4i){(int)d4[0], (int)d4[1], (int)d4[2], (int)d4[3]};
      }
      const int ua = u % KA;
#pragma unroll
      for (int mf = 0; mf < MF2; ++mf) {
        // lanes l and l + 32 of a row exchange halves: v_permlane32_swap swaps the upper 32 lanes of its first operand (h = 1's
        // bytes 16..23) with the lower 32 lanes of its second (h = 0's bytes 8..15)
        v4i bop = aq_[ua][mf];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3"
                     : "+v"(bop[0]), "+v"(bop[1]), "+v"(bop[2]), "+v"(bop[3]));
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mf][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wop[t], bop, acc[mf][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_w(ks0 + u + KD, wq_[u], sq_[u]);
      load_a(ks0 + u + KA, aq_[ua]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- K-split reduction in passes of TPP tiles (128 KiB of LDS: 4 tiles of 8 waves, 2 of 16; tile (mf, t) belongs to wave
  // (4 mf + t) % KS) and epilogue
  extern __shared__ __attribute__((aligned(16))) int red[];  // [KS][TPP tiles][4 register groups][64 lanes] x v4i = 128 KiB
  v4i* red4 = reinterpret_cast<v4i*>(red);
  constexpr int TPP = 32 / KS;
#pragma unroll
  for (int mf = 0; mf < MF2; ++mf) {
#pragma unroll
    for (int t0 = 0; t0 < 4; t0 += TPP) {
      if (mf > 0 || t0 > 0) __syncthreads();  // (the owners of the pass before are done reading)
#pragma unroll
      for (int tt = 0; tt < TPP; ++tt) {
        const int t = t0 + tt;
        if ((mf * 4 + t) % KS != wave) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            red4[((wave * TPP + tt) * 4 + q) * 64 + lane] =
                (v4i){acc[mf][t][4 * q], acc[mf][t][4 * q + 1], acc[mf][t][4 * q + 2], acc[mf][t][4 * q + 3]};
        }
      }
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < TPP; ++tt) {
        const int t = t0 + tt;
        if ((mf * 4 + t) % KS != wave) continue;
        const int m = m0 + mf * 32 + i32;
        const float sa = m < M ? (float)ascales[m] : 0.f;
        const float asum = (GROUP || m >= M) ? 0.f : (float)a_ssums[m];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // register group q = column block q of the quad
          v4i sum = (v4i){acc[mf][t][4 * q], acc[mf][t][4 * q + 1], acc[mf][t][4 * q + 2], acc[mf][t][4 * q + 3]};
#pragma unroll
          for (int w2 = 0; w2 < KS; ++w2)
            if (w2 != wave) {
              const v4i o = red4[((w2 * TPP + tt) * 4 + q) * 64 + lane];
              sum[0] += o[0]; sum[1] += o[1]; sum[2] += o[2]; sum[3] += o[3];
            }
          const int n = (quad * 4 + q) * 32 + t * 8 + h * 4;
          if (m >= M || n >= N) continue;
          Vec<f16, 4> o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = (float)sum[r] * sa * (float)wscales[n + r];
            if constexpr (!GROUP) v -= asum * (float)w_szs[n + r];
            o[r] = (f16)v;
          }
          store_vec<f16, 4>(out + (int64_t)m * ldc + n, o);
        }
      }
    }
  }
}

// Many rows (M > 64): a 128 x 128 tile per workgroup, 4 waves as 2 (m) x 2 (n), a wave owns 64 rows x two 32-column
// weight blocks = 4 x 4 MFMA tiles. The activation tile [128 rows][64 B] of a k step is staged ONCE per workgroup in LDS
// (two 64-deep steps per stage: global -> registers two stages ahead -> LDS one stage ahead, two buffers, one LDS-only
// barrier per 128 k) instead of being
// pulled through the vector L1 by every wave: per 16 MFMAs a wave now reads 4 KiB of activations from LDS and 2 KiB of
// weights from global memory, against 8 KiB + 1 KiB through the L1 before. LDS image: row r at 64 r, its four 16-byte
// parts at (part ^ swz[(r >> 2) & 3]) with swz = {0, 3, 2, 1}: a ds_read_b128 serves lanes {0-3, 12-15, 20-27} together
// (MI355X_MICROARCH.md, LDS), i.e. rows r, r + 12 of lane group kg with rows r + 4, r + 8 of group kg + 1 - four rows
// on the same 16 banks that this swizzle spreads over the four parts.
template <bool GROUP>
__global__ __launch_bounds__(256, 2) void qserve_w4a8_tile_kernel(
    f16* __restrict__ out, const int8_t* __restrict__ a, const uint8_t* __restrict__ w,
    const int8_t* __restrict__ zeros, const int8_t* __restrict__ scales_i8, const f16* __restrict__ wscales,
    const f16* __restrict__ ascales, const f16* __restrict__ w_szs, const f16* __restrict__ a_ssums, int M, int N,
    int K, int64_t lda, int64_t ldc) {
  __shared__ __attribute__((aligned(1024))) char as[2][2][128 * 64];  // [stage buffer][k step of the stage]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 128;
  const int j = lane & 15, kg = lane >> 4;
  const int c = j & 7, b = j >> 3;

  // weights: the wave's two 32-column blocks (clamped: a block past N reads block 0 and is not stored)
  const uint8_t* wl[2];
  const int8_t *s8[2], *z8[2];
  int n32[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    n32[q] = blockIdx.x * 4 + wn * 2 + q;
    const int nc = n32[q] * 32 < N ? n32[q] : 0;
    wl[q] = w + ((int64_t)nc * (K >> 5) + (kg >> 1)) * 512 + c * 64 + (kg & 1) * 8 + b * 4;
    s8[q] = GROUP ? scales_i8 + nc * 32 + c * 4 + b : nullptr;
    z8[q] = GROUP ? zeros + nc * 32 + c * 4 + b : nullptr;
  }
  // activation staging: 16-byte chunk ch = tid + 256 i of the [128][4] tile: row ch / 4, part ch % 4
  const int8_t* ap[2];
  int aoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = tid + 256 * i, row = ch >> 2, part = ch & 3;
    int m = m0 + row;
    m = m < M ? m : M - 1;
    ap[i] = a + (int64_t)m * lda + part * 16;
    aoff[i] = row * 64 + ((part ^ ((4 - ((row >> 2) & 3)) & 3)) << 4);
  }
  // fragment reads: row wm * 64 + 16 mf + j, part kg
  const int rd = (wm * 64 + j) * 64 + ((kg ^ ((4 - ((j >> 2) & 3)) & 3)) << 4);  // (+ 1024 mf: (16 mf) >> 2 is a multiple of 4)

  v4i acc[4][4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (v4i){0, 0, 0, 0};

  const int nks = K >> 6;
  auto load_w = [&](int ks, uint32_t (&wd)[2][4], uint32_t (&sz)[2][4]) {
    ks = ks < nks ? ks : nks - 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) wd[q][e] = *reinterpret_cast<const uint32_t*>(wl[q] + (int64_t)ks * 1024 + e * 16);
      if constexpr (GROUP) {
        const int64_t g = ks >> 1;
        sz[q][0] = (uint8_t)s8[q][g * N]; sz[q][1] = (uint8_t)s8[q][g * N + 2];
        sz[q][2] = (uint8_t)z8[q][g * N]; sz[q][3] = (uint8_t)z8[q][g * N + 2];
      }
    }
  };
  // a stage = two 64-deep k steps: one barrier per 128 k
  auto load_a = [&](int st, v4i (&r)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ks = 2 * st + h;
      ks = ks < nks ? ks : nks - 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) r[2 * h + i] = *reinterpret_cast<const v4i*>(ap[i] + ks * 64);
    }
  };
  auto store_a = [&](int buf, const v4i (&r)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<v4i*>(&as[buf][h][aoff[i]]) = r[2 * h + i];
  };
  // (weights four steps ahead instead of two measured the same: 153 us at 4096^3)
  constexpr int kD = 2;
  uint32_t wq_[kD][2][4], sq_[kD][2][4];
  v4i aq_[4];
  load_a(0, aq_);
  store_a(0, aq_);
  load_a(1, aq_);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int d = 0; d < kD; ++d) {
    load_w(d, wq_[d], sq_[d]);
    __builtin_amdgcn_sched_barrier(0);
  }

  for (int ks0 = 0; ks0 < nks; ks0 += kD) {
#pragma unroll
    for (int u = 0; u < kD; ++u) {
      const int ks = ks0 + u;
      const int st = ks >> 1, buf = st & 1;
      if ((u & 1) == 0) {
        // stage st is in LDS, everyone is done reading the other buffer (LDS-only barrier: the loads in flight stay there)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        store_a(buf ^ 1, aq_);
        load_a(st + 2, aq_);
      }
      const uint32_t keep = ks < nks ? 0xffffffffu : 0u;  // (a step past K multiplies zero weights: no branch)
      // the step's four activation fragments are requested in front of the weight arithmetic (read one at a time in front
      // of its MFMAs, each read exposed the LDS latency: four times per step)
      v4i afq[4];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) afq[mf] = *reinterpret_cast<const v4i*>(&as[buf][u & 1][rd + mf * 1024]);
      __builtin_amdgcn_sched_barrier(0);
      v4i wop[4];                                          // [2 q + half]
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t wd = wq_[u][q][e] & keep;
          uint32_t lo = wd & 0x0f0f0f0fu, hi = (wd >> 4) & 0x0f0f0f0fu;
          if constexpr (GROUP) {
            lo = add_bytes(lo * sq_[u][q][0], (sq_[u][q][2] * 0x01010101u) & keep);
            hi = add_bytes(hi * sq_[u][q][1], (sq_[u][q][3] * 0x01010101u) & keep);
          }
          wop[2 * q][e] = (int)lo;
          wop[2 * q + 1][e] = (int)hi;
        }
      }
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wop[nf], afq[mf], acc[mf][nf], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_w(ks + kD, wq_[u], sq_[u]);  // (after the slot's last use: see the kernel above)
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: lane (column m = j of the m fragment, rows n = 4 kg + r of the n fragment)
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    const int m = m0 + wm * 64 + mf * 16 + j;
    if (m >= M) continue;
    const float sa = (float)ascales[m];
    const float asum = GROUP ? 0.f : (float)a_ssums[m];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int n = n32[nf >> 1] * 32 + (nf & 1) * 16 + kg * 4;
      if (n >= N) continue;
      Vec<f16, 4> o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (float)acc[mf][nf][r] * sa * (float)wscales[n + r];
        if constexpr (!GROUP) v -= asum * (float)w_szs[n + r];
        o[r] = (f16)v;
      }
      store_vec<f16, 4>(out + (int64_t)m * ldc + n, o);
    }
  }
}

#ifdef SGLK_PROBES
static int g_qserve_mf = 0;  // (diagnostic: cap on the m-tiles per workgroup of the decode stream kernel)
static int g_qserve_cfg = 0; // (diagnostic: forced stream configuration 10000 mf + 1000 log2(ks) + 10 kd + ka)
#else
constexpr int g_qserve_mf = 0;
constexpr int g_qserve_cfg = 0;
#endif
#ifdef SGLK_PROBES
static int g_qserve_persist_rows = 128;
#define kPersistRows g_qserve_persist_rows
#else
constexpr int kPersistRows = 128;  // above this many rows: the persistent pipeline of gemm_8bit.hip
#endif

template <bool GROUP>
static int launch(hipStream_t st, void* out, const void* a, const void* w, const void* zeros, const void* scales_i8,
                  const void* wscales, const void* ascales, const void* w_szs, const void* a_ssums, int64_t M, int64_t N,
                  int64_t K, int64_t lda, int64_t ldc) {
  const unsigned gx = (unsigned)cdiv(N, 128);
  // few rows: the weight stream. K is split over KS waves of a workgroup until the launch has ~1500 waves (a wave keeps
  // at least four 64-deep steps); 16 waves (128 registers per lane) only with one m-tile and a 4-deep ring
  const int64_t pairs = cdiv(N, 64);
  // m-tiles per workgroup: 1 / 2 / 4 by the row count, fewer (the weights are then streamed once per group of m-tiles, the
  // repeats from L2) when the launch would have under 256 workgroups - N = K = 4096, 64 rows: 24.9 us with 4 m-tiles on 64
  // workgroups, 16.8 with 2 on 128, 13.0 with 1 on 256; at N = 14336 one m-tile per workgroup is slower (30.0 against 26.2 us)
  int mf = M <= 16 ? 1 : M <= 32 ? 2 : 4;
  // (round 5, with the 16-byte activation loads: N = 14336 keeps four m-tiles on 224 workgroups - 14.8 us at 64 rows against 19 with
  //  two m-tiles on 448; N = 4096 goes down to one m-tile on 256 workgroups - 8.8 against 14.0 us)
  while (mf > 1 && pairs * cdiv(M, 16 * mf) < 192) mf /= 2;
  if (g_qserve_mf > 0 && g_qserve_mf < mf) mf = g_qserve_mf;
  int ks = 1;
  while (ks < (M <= 16 ? 16 : 8) && pairs * cdiv(M, 16 * mf) * ks < 1536 && (K >> 6) >= 8 * ks) ks *= 2;
  const bool deep = (K >> 6) / ks >= 8 && ks <= 8;  // (ring no deeper than a wave's steps)
#define SGLK_GO_STREAM(MF, KS, KD, KA) SGLK_GO_STREAM_IL(MF, KS, KD, KA, true)
#define SGLK_GO_STREAM_IL(MF, KS, KD, KA, IL)                                                                    \
  {                                                                                                              \
    constexpr int lds = KS > 1 ? KS * MF * 4 * 64 * 16 : 0;                                                      \
    static unsigned long long attr_done = 0;                                                                     \
    if (lds > 64 * 1024)                                                                                         \
      if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&qserve_w4a8_stream_kernel<GROUP, MF, KS, KD, KA, IL>), lds, \
                                   &attr_done, "qserve_w4a8"))                                                   \
        return rc;                                                                                               \
    qserve_w4a8_stream_kernel<GROUP, MF, KS, KD, KA, IL><<<dim3((unsigned)pairs, (unsigned)cdiv(M, 16 * MF)), 64 * KS, lds, st>>>( \
        (f16*)out, (const int8_t*)a, (const uint8_t*)w, (const int8_t*)zeros, (const int8_t*)scales_i8,          \
        (const f16*)wscales, (const f16*)ascales, (const f16*)w_szs, (const f16*)a_ssums, (int)M, (int)N, (int)K, lda, ldc); \
  }
#define SGLK_GO_STREAM_KS(MF, KD, KA)                                                                            \
  {                                                                                                              \
    if (ks == 1) SGLK_GO_STREAM(MF, 1, KD, KA) else if (ks == 2) SGLK_GO_STREAM(MF, 2, KD, KA)                   \
    else if (ks == 4) SGLK_GO_STREAM(MF, 4, KD, KA) else SGLK_GO_STREAM(MF, 8, KD, KA)                           \
  }
#define SGLK_GO_STREAM32(MF2, KD, KA) SGLK_GO_STREAM32_KS(MF2, KD, KA, 8)
#define SGLK_GO_STREAM32_KS(MF2, KD, KA, KS)                                                                     \
  {                                                                                                              \
    constexpr int lds = 8 * 4 * 4 * 64 * 16;                                                                     \
    static unsigned long long attr_done = 0;                                                                     \
    if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&qserve_w4a8_stream32_kernel<GROUP, MF2, KD, KA, KS>), lds, &attr_done, \
                                 "qserve_w4a8"))                                                                 \
      return rc;                                                                                                 \
    qserve_w4a8_stream32_kernel<GROUP, MF2, KD, KA, KS><<<dim3((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 32 * MF2)), 64 * KS, lds, st>>>( \
        (f16*)out, (const int8_t*)a, (const uint8_t*)w, (const int8_t*)zeros, (const int8_t*)scales_i8,          \
        (const f16*)wscales, (const f16*)ascales, (const f16*)w_szs, (const f16*)a_ssums, (int)M, (int)N, (int)K, lda, ldc); \
  }
#ifdef SGLK_PROBES
  if (g_qserve_cfg > 1 && (M <= 64 || (g_qserve_cfg >= 300000 && M <= 512))) {  // forced configuration: 10000 mf + 1000 log2(ks) + 10 kd + ka (1: the stream kernel where the 32x32x32 form is the default)
    switch (g_qserve_cfg) {
      case 43082: SGLK_GO_STREAM(4, 8, 8, 2) break;
      case 43042: SGLK_GO_STREAM(4, 8, 4, 2) break;
      case 43022: SGLK_GO_STREAM(4, 8, 2, 2) break;
      case 143022: SGLK_GO_STREAM_IL(4, 8, 2, 2, false) break;
      case 123044: SGLK_GO_STREAM_IL(2, 8, 4, 4, false) break;
      case 43084: SGLK_GO_STREAM(4, 8, 8, 4) break;
      case 42082: SGLK_GO_STREAM(4, 4, 8, 2) break;
      case 23084: SGLK_GO_STREAM(2, 8, 8, 4) break;
      case 23044: SGLK_GO_STREAM(2, 8, 4, 4) break;
      case 23082: SGLK_GO_STREAM(2, 8, 8, 2) break;
      case 24044: SGLK_GO_STREAM(2, 16, 4, 4) break;
      case 13088: SGLK_GO_STREAM(1, 8, 8, 8) break;
      case 13084: SGLK_GO_STREAM(1, 8, 8, 4) break;
      case 14044: SGLK_GO_STREAM(1, 16, 4, 4) break;
      case 14042: SGLK_GO_STREAM(1, 16, 4, 2) break;
      case 320022: SGLK_GO_STREAM32(2, 2, 2) break;
      case 320042: SGLK_GO_STREAM32(2, 4, 2) break;
      case 320044: SGLK_GO_STREAM32(2, 4, 4) break;
      case 320082: SGLK_GO_STREAM32(2, 8, 2) break;
      case 320084: SGLK_GO_STREAM32(2, 8, 4) break;
      case 311011: SGLK_GO_STREAM32_KS(1, 1, 1, 16) break;
      case 311022: SGLK_GO_STREAM32_KS(1, 2, 2, 16) break;
      case 311021: SGLK_GO_STREAM32_KS(1, 2, 1, 16) break;
      case 310022: SGLK_GO_STREAM32(1, 2, 2) break;
      case 310021: SGLK_GO_STREAM32(1, 2, 1) break;
      case 310011: SGLK_GO_STREAM32(1, 1, 1) break;
      case 310042: SGLK_GO_STREAM32(1, 4, 2) break;
      case 310044: SGLK_GO_STREAM32(1, 4, 4) break;
      case 310084: SGLK_GO_STREAM32(1, 8, 4) break;
      default: return SGLK_OK;
    }
    return check_launch("qserve_w4a8(cfg)");
  }
#endif
  // 33 - 64 rows at wide N (round 5): the 32x32x32 form, one 32-row m-tile per workgroup - at 14336 x 4096 and 64 rows 12.4 us
  // per channel / 13.4 per group against 14.5 / 15.1 for the stream above (48 rows: 12.1 / 13.1 against 13.2 / 13.5); two m-tiles
  // per workgroup (half the workgroups, the weights streamed once) lose to it - 18 us: 112 CUs then carry all the int8 MFMAs;
  // up to 32 rows and at narrow N (under 192 workgroups) the stream above stays faster (N = 4096, 64 rows: 8.6 against 11.6 us)
  // 65 - 512 rows (round 5, late): the same kernel with M / 32 (one m-tile per workgroup) or M / 64 (two) workgroup rows, each
  // streaming the weights (the repeats come from L2 / the Infinity Cache). Before, 65 - 128 rows ran on the 128 x 128 tile kernel
  // below and 129+ on the persistent pipeline, whose floor is ~38 - 55 us whatever the size: 14336 x 4096 at 65 / 128 / 256 rows
  // 39 - 56 -> 19.3 / 19.8 / 37 us, 4096 x 4096 at 65 / 128 / 256 / 512 rows 38 - 53 -> 11.3 / 11.6 / 12.3 / 19.7 us, outputs
  // bit-identical. A round of one-m-tile workgroups takes ~12 us, of two-m-tile ones ~19 us (K = 4096); the form with the smaller
  // estimate runs when it beats the other paths' floor - always up to 128 rows, up to ~36 us above.
  bool stream32 = false, two_tiles = false;
  if (M > 64 && M <= 1024 && g_qserve_cfg != 1) {
    const int64_t quads = cdiv(N, 128), cus = num_cus() > 0 ? num_cus() : 256;
    const int64_t e1 = cdiv(quads * cdiv(M, 32), cus) * 12, e2 = cdiv(quads * cdiv(M, 64), cus) * 19;
    // (round 5, late: both sides of the comparison scale with K. The rule compared the stream's estimate, scaled by K / 4096, with the
    //  pipeline's floor AT K = 4096 - so at the down projection's shape, N = 4096, K = 14336, everything from 129 rows on went to the
    //  pipeline: 125 us per channel / 156 per group whatever the rows, against 30 at 128 rows. The pipeline runs a K block of a half
    //  tile in ~1.12 us per channel, ~1.4 per group, one round per 256 half tiles (128 x 256; whole tiles above 512 rows unless they
    //  fill at most half of the CUs).)
    const double kscale = (double)K / 4096.0;  // (the stream estimates are for K = 4096)
    const int64_t tiles = cdiv(M, 256) * cdiv(N, 256);
    const int64_t units = (M <= 512 || tiles <= cus / 2) ? 2 * tiles : tiles;
    const double pipe = (double)cdiv(units, cus) * (double)(K / 128) * (GROUP ? 1.4 : 1.12) * (units == tiles ? 1.6 : 1.0);
    // (0.8: the stream's repeats come from L2 and run under their estimates - N = 4096, K = 14336 at 256 / 512 rows 30 / 48 us against
    //  estimates of 42 / 66; at a tie the stream wins - 8192^2 at 257 rows 58 against 75 us, N = 14336, K = 4096 at 256 rows 37 against 44)
    stream32 = M <= 128 || 0.8 * (double)std::min(e1, e2) * kscale <= pipe;
    two_tiles = e2 < e1;
  }
  if (stream32) {
    if (two_tiles) SGLK_GO_STREAM32(2, 2, 2) else SGLK_GO_STREAM32(1, 1, 1)
  } else if (M > 40 && M <= 64 && cdiv(N, 128) * cdiv(M, 32) >= 192 && g_qserve_cfg != 1) {
    SGLK_GO_STREAM32(1, 1, 1)
  } else if (M <= 64) {
    if (mf == 1) {
      // (ring depths from the round-5 sweep at N = 4096 / 14336, K = 4096, 1 - 64 rows: shallow activation rings win everywhere, a
      //  weight ring over a wave's whole k range does not - 16.1 us at (4 m-tiles, 8 waves, 8, 2) against 14.8 at (4, 8, 2, 2))
      if (ks == 16) SGLK_GO_STREAM(1, 16, 4, 2) else if (deep) SGLK_GO_STREAM_KS(1, 8, 4) else SGLK_GO_STREAM_KS(1, 4, 2)
    } else if (mf == 2) SGLK_GO_STREAM_KS(2, 4, 4)
    else SGLK_GO_STREAM_KS(4, 2, 2)
  } else if (M > kPersistRows && qserve_w4a8_persist(st, GROUP, out, a, w, zeros, scales_i8, wscales, ascales, w_szs, a_ssums, M, N, K,
                                                   lda, ldc)) {
  } else {
    qserve_w4a8_tile_kernel<GROUP><<<dim3(gx, (unsigned)cdiv(M, 128)), 256, 0, st>>>(
        (f16*)out, (const int8_t*)a, (const uint8_t*)w, (const int8_t*)zeros, (const int8_t*)scales_i8,
        (const f16*)wscales, (const f16*)ascales, (const f16*)w_szs, (const f16*)a_ssums, (int)M, (int)N, (int)K, lda, ldc);
  }
#undef SGLK_GO_STREAM_KS
#undef SGLK_GO_STREAM32
#undef SGLK_GO_STREAM32_KS
#undef SGLK_GO_STREAM
#undef SGLK_GO_STREAM_IL
  return check_launch(GROUP ? "qserve_w4a8_per_group_gemm" : "qserve_w4a8_per_chn_gemm");
}

static int check(const char* op, const void* out, const void* a, const void* w, int64_t M, int64_t N, int64_t K,
                 int64_t lda, int64_t ldc, int64_t kmult) {
  SGLK_REQUIRE(M >= 0 && N > 0 && K > 0, "%s: bad shape M=%lld N=%lld K=%lld", op, (long long)M, (long long)N, (long long)K);
  SGLK_REQUIRE(N % 32 == 0, "%s: N=%lld must be a multiple of 32 (QServe 32x32 weight blocks)", op, (long long)N);
  SGLK_REQUIRE(K % kmult == 0, "%s: K=%lld must be a multiple of %lld", op, (long long)K, (long long)kmult);
  SGLK_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "%s: shape too large", op);
  SGLK_REQUIRE(lda % 16 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)w % 16 == 0,
               "%s: activation rows and the packed weight must be 16-byte aligned", op);
  SGLK_REQUIRE(ldc % 4 == 0 && (uintptr_t)out % 8 == 0, "%s: output rows must be 8-byte aligned", op);
  return SGLK_OK;
}

}  // namespace
}  // namespace sglk

#ifdef SGLK_PROBES
extern "C" SGLK_API void sglk_debug_set_qserve_persist_rows(int rows) { sglk::g_qserve_persist_rows = rows; }
extern "C" SGLK_API void sglk_debug_set_qserve_mf(int mf) { sglk::g_qserve_mf = mf; }
extern "C" SGLK_API void sglk_debug_set_qserve_cfg(int cfg) { sglk::g_qserve_cfg = cfg; }
#endif

extern "C" int sglk_qserve_w4a8_per_chn_gemm(sglk_stream_t stream, void* out, const void* in_feats, const void* kernel,
                                             const void* wscales, const void* ascales, const void* w_szs,
                                             const void* a_ssums, int64_t M, int64_t N, int64_t K, int64_t lda,
                                             int64_t ldc) {
  using namespace sglk;
  if (int rc = check("qserve_w4a8_per_chn_gemm", out, in_feats, kernel, M, N, K, lda, ldc, 64)) return rc;
  if (M == 0) return SGLK_OK;
  return launch<false>((hipStream_t)stream, out, in_feats, kernel, nullptr, nullptr, wscales, ascales, w_szs, a_ssums, M, N,
                       K, lda, ldc);
}

extern "C" int sglk_qserve_w4a8_per_group_gemm(sglk_stream_t stream, void* out, const void* in_feats, const void* kernel,
                                               const void* zeros, const void* scales_i8, const void* wscales,
                                               const void* ascales, int64_t M, int64_t N, int64_t K, int64_t lda,
                                               int64_t ldc) {
  using namespace sglk;
  if (int rc = check("qserve_w4a8_per_group_gemm", out, in_feats, kernel, M, N, K, lda, ldc, 128)) return rc;
  // (the decode kernel reads a lane's four group scales / zero terms as one dword)
  SGLK_REQUIRE((uintptr_t)scales_i8 % 4 == 0 && (uintptr_t)zeros % 4 == 0,
               "qserve_w4a8_per_group_gemm: scales_i8 and zeros must be 4-byte aligned");
  if (M == 0) return SGLK_OK;
  return launch<true>((hipStream_t)stream, out, in_feats, kernel, zeros, scales_i8, wscales, ascales, nullptr, nullptr, M, N,
                      K, lda, ldc);
}
