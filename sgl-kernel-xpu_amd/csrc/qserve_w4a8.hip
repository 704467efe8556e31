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
// No LDS, no barriers: a wave owns a 32-column block of W for all of K, 16*MF rows of A; loads run 2-4 k steps ahead.
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

// SPLIT = 1: workgroup = 4 waves on 4 neighbouring 32-column blocks, each over all of K.
// SPLIT = 16 / 8 (few rows, M <= 16 / 64: the weight stream of a decode step): workgroup = SPLIT waves on ONE 32-column
// block, wave w takes the 64-deep k steps w, w + SPLIT, ..; the int32 partial sums are added through LDS (exact) and wave
// 0 finishes (8 waves above 16 rows: 16 would leave 128 registers per lane and spill).
// With one wave per 32 columns over all of K a 4096 x 4096 layer is 128 waves of 64 dependent steps each: 19 us.
template <bool GROUP, int MF, int SPLIT>
__global__ __launch_bounds__(SPLIT == 1 ? 256 : 64 * SPLIT) void qserve_w4a8_kernel(
    f16* __restrict__ out, const int8_t* __restrict__ a, const uint8_t* __restrict__ w,
    const int8_t* __restrict__ zeros, const int8_t* __restrict__ scales_i8, const f16* __restrict__ wscales,
    const f16* __restrict__ ascales, const f16* __restrict__ w_szs, const f16* __restrict__ a_ssums, int M, int N,
    int K, int64_t lda, int64_t ldc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n32 = SPLIT == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  if (SPLIT == 1 && n32 * 32 >= N) return;
  const int m0 = blockIdx.y * (16 * MF);
  const int j = lane & 15, kg = lane >> 4;
  const int c = j & 7, b = j >> 3;

  // weights: block (n32, k32) at (n32 * K/32 + k32) * 512; this lane's dword e of k step ks (64 deep): k32 = 2 ks + kg/2
  const uint8_t* wl = w + ((int64_t)n32 * (K >> 5) + (kg >> 1)) * 512 + c * 64 + (kg & 1) * 8 + b * 4;
  // activations: row m0 + 16 mf + j (clamped), 16 bytes at k = 64 ks + 16 kg
  const int8_t* al[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    int m = m0 + mf * 16 + j;
    m = m < M ? m : M - 1;
    al[mf] = a + (int64_t)m * lda + kg * 16;
  }
  // group scales / zero terms of this lane's two rows (low nibble: column 8 b + c -> position 4 c + b; high: 4 c + 2 + b)
  const int8_t* s8 = GROUP ? scales_i8 + n32 * 32 + c * 4 + b : nullptr;
  const int8_t* z8 = GROUP ? zeros + n32 * 32 + c * 4 + b : nullptr;

  v4i acc[MF][2];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) acc[mf][0] = acc[mf][1] = (v4i){0, 0, 0, 0};

  // Weights, activations and group scales run kD 64-deep steps ahead of the MFMAs in a register ring with static slots
  // (loop unrolled kD times): a load consumed in the iteration that issues it exposes the whole memory latency per step.
  // Steps past K re-read the last one (their results are not accumulated).
  const int nks_all = K >> 6;
  // this wave's steps: ks = kfirst + kstride * i, i < nks
  const int kfirst = SPLIT == 1 ? 0 : wave, kstride = SPLIT;
  const int nks = SPLIT == 1 ? nks_all : (nks_all > wave ? (nks_all - wave + SPLIT - 1) / SPLIT : 0);
  constexpr int kD = MF <= 2 ? 4 : 2;
  uint32_t wq_[kD][4], sq_[kD][4];
  v4i aq_[kD][MF];
  auto load_step = [&](int i, uint32_t (&wd)[4], uint32_t (&sz)[4], v4i (&af)[MF]) {
    i = i < nks ? i : nks - 1;
    int ks = kfirst + kstride * (i > 0 ? i : 0);
    ks = ks < nks_all ? ks : nks_all - 1;  // (a wave without steps loads the last one; nothing is accumulated)
#pragma unroll
    for (int e = 0; e < 4; ++e) wd[e] = *reinterpret_cast<const uint32_t*>(wl + (int64_t)ks * 1024 + e * 16);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) af[mf] = *reinterpret_cast<const v4i*>(al[mf] + ks * 64);
    if constexpr (GROUP) {
      const int64_t g = ks >> 1;
      sz[0] = (uint8_t)s8[g * N]; sz[1] = (uint8_t)s8[g * N + 2];
      sz[2] = (uint8_t)z8[g * N]; sz[3] = (uint8_t)z8[g * N + 2];
    }
  };
#pragma unroll
  for (int d = 0; d < kD; ++d) load_step(d, wq_[d], sq_[d], aq_[d]);
  for (int ks0 = 0; ks0 < nks; ks0 += kD) {
#pragma unroll
    for (int u = 0; u < kD; ++u) {
      // (steps past this wave's range multiply zero weights: no branch; the slot is refilled AFTER its last use, else the
      // new step lives in other registers and the loop end moves the ring back with copies that wait for every load)
      const uint32_t keep = ks0 + u < nks ? 0xffffffffu : 0u;
      v4i wlo, whi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t wd = wq_[u][e] & keep;
        uint32_t lo = wd & 0x0f0f0f0fu, hi = (wd >> 4) & 0x0f0f0f0fu;
        if constexpr (GROUP) {
          // code * scale <= 15 * 17 fits a byte: one 32-bit multiply scales four codes; the add wraps per byte
          lo = add_bytes(lo * sq_[u][0], (sq_[u][2] * 0x01010101u) & keep);
          hi = add_bytes(hi * sq_[u][1], (sq_[u][3] * 0x01010101u) & keep);
        }
        wlo[e] = (int)lo;
        whi[e] = (int)hi;
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        acc[mf][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wlo, aq_[u][mf], acc[mf][0], 0, 0, 0);
        acc[mf][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(whi, aq_[u][mf], acc[mf][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_step(ks0 + u + kD, wq_[u], sq_[u], aq_[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if constexpr (SPLIT > 1) {
    // (dword r of a lane at stride 64: every LDS access is 64 consecutive dwords - the 16-byte form had a bank-conflict ratio of
    // 0.75 in the r02 profile)
    extern __shared__ int red[];  // [SPLIT - 1][MF][2][4][64 lanes]
    if (wave > 0) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(((((wave - 1) * MF + mf) * 2 + h) * 4 + r) << 6) + lane] = acc[mf][h][r];
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 0; w < SPLIT - 1; ++w) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mf][h][r] += red[((((w * MF + mf) * 2 + h) * 4 + r) << 6) + lane];
        }
    }
  }

  // ---- epilogue: lane (column m = j of the m fragment, rows n = 4 kg + r of the n fragment)
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = m0 + mf * 16 + j;
    if (m >= M) continue;
    const float sa = (float)ascales[m];
    const float asum = GROUP ? 0.f : (float)a_ssums[m];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = n32 * 32 + h * 16 + kg * 4;
      Vec<f16, 4> o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (float)acc[mf][h][r] * sa * (float)wscales[n + r];
        if constexpr (!GROUP) v -= asum * (float)w_szs[n + r];
        o[r] = (f16)v;
      }
      store_vec<f16, 4>(out + (int64_t)m * ldc + n, o);
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
#define SGLK_GO(MF)                                                                                              \
  qserve_w4a8_kernel<GROUP, MF, 1><<<dim3(gx, (unsigned)cdiv(M, 16 * MF)), 256, 0, st>>>(                        \
      (f16*)out, (const int8_t*)a, (const uint8_t*)w, (const int8_t*)zeros, (const int8_t*)scales_i8,            \
      (const f16*)wscales, (const f16*)ascales, (const f16*)w_szs, (const f16*)a_ssums, (int)M, (int)N, (int)K, lda, ldc)
#define SGLK_GO_SPLIT(MF, SPLIT)                                                                                 \
  {                                                                                                              \
    constexpr int lds = (SPLIT - 1) * MF * 2 * 64 * 16;                                                          \
    static unsigned long long attr_done = 0;                                                                     \
    if (lds > 64 * 1024)                                                                                         \
      if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&qserve_w4a8_kernel<GROUP, MF, SPLIT>), lds, &attr_done, \
                                   "qserve_w4a8"))                                                               \
        return rc;                                                                                               \
    qserve_w4a8_kernel<GROUP, MF, SPLIT><<<dim3((unsigned)(N / 32), (unsigned)cdiv(M, 16 * MF)), 64 * SPLIT, lds, st>>>( \
        (f16*)out, (const int8_t*)a, (const uint8_t*)w, (const int8_t*)zeros, (const int8_t*)scales_i8,          \
        (const f16*)wscales, (const f16*)ascales, (const f16*)w_szs, (const f16*)a_ssums, (int)M, (int)N, (int)K, lda, ldc); \
  }
  if (M <= 16) SGLK_GO_SPLIT(1, 16)
  else if (M <= 32) SGLK_GO_SPLIT(2, 8)
  else if (M <= 64) SGLK_GO_SPLIT(4, 8)
  else if (M > kPersistRows && qserve_w4a8_persist(st, GROUP, out, a, w, zeros, scales_i8, wscales, ascales, w_szs, a_ssums, M, N, K,
                                                   lda, ldc)) {
  } else {
    qserve_w4a8_tile_kernel<GROUP><<<dim3(gx, (unsigned)cdiv(M, 128)), 256, 0, st>>>(
        (f16*)out, (const int8_t*)a, (const uint8_t*)w, (const int8_t*)zeros, (const int8_t*)scales_i8,
        (const f16*)wscales, (const f16*)ascales, (const f16*)w_szs, (const f16*)a_ssums, (int)M, (int)N, (int)K, lda, ldc);
  }
#undef SGLK_GO
#undef SGLK_GO_SPLIT
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
  if (M == 0) return SGLK_OK;
  return launch<true>((hipStream_t)stream, out, in_feats, kernel, zeros, scales_i8, wscales, ascales, nullptr, nullptr, M, N,
                      K, lda, ldc);
}
