// 8-bit scaled GEMMs for gfx950: fp8_blockwise_scaled_mm (DeepSeek-style 1x128 /
// 128x128 block scales), fp8_scaled_mm and int8_scaled_mm (per-token x per-channel).
//
// ---- fp8_blockwise_scaled_mm ----
// The reference only declares this op (include/sgl_kernel_ops.h:581-586) and
// pins its meaning with tests/test_fp8_blockwise_gemm.py:23-85:
//   out = T( (sa (x) a) @ (sb (x) b) ),  a [M,K] e4m3 row-major, b [K,N] e4m3
//   column-major, sa [M,K/128], sb [K/128,N/128] fp32.
// There is no reference kernel; this one is designed for CDNA4 from scratch.
//
// Arithmetic: for each 128-deep K block kb the 8-bit products are summed by
// ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16x16 output tile (unit E8M0
// hardware scales: this form runs at twice the rate of the non-MX fp8 MFMA),
// into a fresh fp32 partial; the block scale is applied on the VALU:
//   acc[m,n] += partial[m,n] * (sa[m,kb] * sb[kb,n/128])
// and acc is rounded once to T at the end.
//
// Tile: 256(M) x 256(N) x 128(K) per 512-thread workgroup, 8 waves as 2(M) x 4(N),
// each wave 128 x 64 = 8 x 4 MFMA tiles (128 accumulator VGPRs).
// Operands are swapped (MFMA "A" = rows of b^T, MFMA "B" = rows of a) so that a
// lane owns 4 consecutive n of one m: one row-scale register per m-fragment and
// 8-byte output stores.
// Staging: a, b^T and the sa column of a K block go global->LDS by LDS-DMA
// (global_load_lds, 16 B/lane), double buffered, one barrier per K block. LDS
// rows are 128 B; the 16-byte chunk c of row r is kept at position
// c ^ ((r>>1)&7) (source-side swizzle, linear DMA destination) which makes both
// ds_read_b128 of a fragment conflict-free: lane (row j, k-group g) reads
// chunks g and g+4. That k order is the same for both operands, so the MFMA
// pairs equal k.
//
// ---- fp8_scaled_mm / int8_scaled_mm ----
// Declared only in the reference (include/sgl_kernel_ops.h:567-580); meaning pinned by
// tests/test_fp8_gemm.py:11-19 and tests/test_int8_gemm.py:16-22. Same tile, staging and
// fragment addressing; the MFMA accumulates across K blocks directly (fp8: the MX K=128
// form; int8: two v_mfma_i32_16x16x64_i8 per block) and the scales are applied once in the
// epilogue:  fp8 : out = T(T(acc * sa[m] * sb[n]) + bias[n])   (bias added in T, after the cast)
//            int8: out = T(float(acc) * sa[m] * sb[n] + bias[n]) (bias added in fp32)
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sglk {
namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 128;
constexpr int kTileBytes = BM * BK;              // 32 KiB per operand per stage
constexpr int kStageBytes = 2 * kTileBytes + 2048;  // + 256 fp32 row scales (+ 1 KiB spare, see pipe kernel)
constexpr int kStages = 2;

#define SGLK_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define SGLK_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ v8i read_frag(const char* tile, int off) {
  const v4i lo = *reinterpret_cast<const v4i*>(tile + off);
  const v4i hi = *reinterpret_cast<const v4i*>(tile + (off ^ 64));
  v8i r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// MODE_W4A8_CHN / MODE_W4A8_GRP: the QServe W4A8 GEMMs (qserve_w4a8.hip) on the persistent pipeline, see below
enum { MODE_BLOCKWISE = 0, MODE_FP8_ROWCOL = 1, MODE_INT8_ROWCOL = 2, MODE_W4A8_CHN = 3, MODE_W4A8_GRP = 4 };

// fp8 e4m3 x e4m3, K = 128, D = A*B + C
// fragment = two 16-byte LDS reads at (addr) and (addr ^ 64); addr is a 32-bit LDS byte address
__device__ __forceinline__ v8i read_frag_lds(uint32_t lo, uint32_t hi, int imm) {
  typedef __attribute__((address_space(3))) const v4i* lds_v4i_ptr;
  const v4i a = *(lds_v4i_ptr)(uintptr_t)(lo + imm);
  const v4i b = *(lds_v4i_ptr)(uintptr_t)(hi + imm);
  v8i r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

template <bool HW_SCALE>
__device__ __forceinline__ v4f mfma_k128(const v8i& a, const v8i& b, const v4f& c) {
  if constexpr (HW_SCALE) {
    // E8M0 127 == 2^0 for both operands
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  } else {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  }
}

// int8 x int8, K = 128 as two K = 64 steps (each operand half is one 16-byte read)
__device__ __forceinline__ v4i mfma_i8_k128(const v8i& a, const v8i& b, v4i c) {
  const v4i a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
  const v4i b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, c, 0, 0, 0);
  return c;
}

template <typename OutT, int MODE, bool VEC_STORE, bool HW_SCALE, int VAR>
__global__ __launch_bounds__(512) void gemm_8bit_kernel(
    OutT* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ sa, const float* __restrict__ sb, const OutT* __restrict__ bias, int M, int N,
    int K, int64_t lda,
    int64_t ldb, int64_t ldc, int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, int tiles_m,
    int tiles_n) {
  __shared__ __attribute__((aligned(256))) char smem[kStages * kStageBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- tile id: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles,
  // walked in groups of 4 m-tiles so that co-resident tiles share a and b panels in L2.
  int tile;
  {
    const int nt = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GM = 4;
  const int group = tile / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int in_group = tile - group * GM * tiles_n;
  const int tm = first_m + in_group % gsz;
  const int tn = in_group / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane source offsets for the LDS-DMA of one K block
  // wave-instruction i of wave w fills rows (4w+i)*8 .. +7 of a tile: lane -> (row, position)
  uint32_t off_a[4], off_b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int ra = (m0 + row < M) ? row : (M - 1 - m0);
    const int rb = (n0 + row < N) ? row : (N - 1 - n0);
    off_a[i] = (uint32_t)((int64_t)ra * lda + chunk * 16);
    off_b[i] = (uint32_t)((int64_t)rb * ldb + chunk * 16);
  }
  const uint8_t* a_tile = a + (int64_t)m0 * lda;
  const uint8_t* b_tile = b + (int64_t)n0 * ldb;
  // row scales: waves 0..3 each fetch 64 of the tile's 256 rows
  const int srow = (m0 + tid < M) ? (m0 + tid) : (M - 1);
  const float* sa_lane = sa + (int64_t)srow * sa_sm;

  auto stage = [&](int kb, int s) {
    char* base = smem + s * kStageBytes;
    const uint8_t* ag = a_tile + (int64_t)kb * BK;
    const uint8_t* bg = b_tile + (int64_t)kb * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds(SGLK_GLB(ag + off_a[i]), SGLK_LDS(base + (wave * 4 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds(SGLK_GLB(bg + off_b[i]),
                                       SGLK_LDS(base + kTileBytes + (wave * 4 + i) * 1024), 16, 0, 0);
    }
    if constexpr (MODE == MODE_BLOCKWISE) {
      if (wave < 4) {
        __builtin_amdgcn_global_load_lds(SGLK_GLB(sa_lane + (int64_t)kb * sa_sk),
                                         SGLK_LDS(base + 2 * kTileBytes + wave * 256), 4, 0, 0);
      }
    }
  };

  // the same DMA split in four parts (two 1-KiB pieces each; part 0 also carries the row scales) so that it can be
  // issued between the MFMA clusters of the running block instead of in one burst behind the barrier
  auto stage_part = [&](int kb, int s, int part) {
    char* base = smem + s * kStageBytes;
    const uint8_t* ag = a_tile + (int64_t)kb * BK;
    const uint8_t* bg = b_tile + (int64_t)kb * BK;
    // per-lane source offsets are recomputed here (a handful of VALU ops) instead of living in registers; the
    // empty asm keeps the compiler from hoisting them out of the K loop (it then spills them: scratch reloads
    // are vector-memory loads and would drain the LDS-DMA queue)
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ii = (part & 1) * 2 + i;
      const int row = (wave * 4 + ii) * 8 + (lane_v >> 3);
      const int chunk = (lane_v & 7) ^ ((row >> 1) & 7);
      if (part < 2) {
        const int ra = (m0 + row < M) ? row : (M - 1 - m0);
        const uint32_t off = (uint32_t)ra * (uint32_t)lda + chunk * 16;
        __builtin_amdgcn_global_load_lds(SGLK_GLB(ag + off), SGLK_LDS(base + (wave * 4 + ii) * 1024), 16, 0, 0);
      } else {
        const int rb = (n0 + row < N) ? row : (N - 1 - n0);
        const uint32_t off = (uint32_t)rb * (uint32_t)ldb + chunk * 16;
        __builtin_amdgcn_global_load_lds(SGLK_GLB(bg + off), SGLK_LDS(base + kTileBytes + (wave * 4 + ii) * 1024), 16, 0, 0);
      }
    }
    if constexpr (MODE == MODE_BLOCKWISE) {
      if (part == 0 && wave < 4) {
        __builtin_amdgcn_global_load_lds(SGLK_GLB(sa_lane + (int64_t)kb * sa_sk),
                                         SGLK_LDS(base + 2 * kTileBytes + wave * 256), 4, 0, 0);
      }
    }
  };

  // ---- fragment addressing (see header): lane = (row j, k-group g)
  const int j = lane & 15, g = lane >> 4;
  const int frag_off = j * 128 + ((g ^ ((j >> 1) & 7)) << 4);
  const int nblk_max = (N + 127) / 128 - 1;
  int nblk = (n0 + wn * 64) >> 7;
  nblk = nblk < nblk_max ? nblk : nblk_max;
  const float* sb_wave = sb + (int64_t)nblk * sb_sn;

  using AccT = typename std::conditional<MODE == MODE_INT8_ROWCOL, v4i, v4f>::type;
  AccT acc[8][4];
#pragma unroll
  for (int mf = 0; mf < 8; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mf][nf][r] = 0;

  const int nkb = K / BK;
  float sbv_next = (MODE == MODE_BLOCKWISE) ? sb_wave[0] : 0.f;
  const uint32_t lds_base = (uint32_t)(uintptr_t)SGLK_LDS(smem);
  stage(0, 0);

  for (int kb = 0; kb < nkb; ++kb) {
    const int s = kb & 1;
    // the DMA of block kb has landed for every wave, and every wave has finished reading stage s^1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (VAR == 0 || VAR == 9 || MODE != MODE_BLOCKWISE) {
      if (kb + 1 < nkb) stage(kb + 1, s ^ 1);
    }
    if constexpr (VAR == 9) continue;  // timing probe: DMA + barriers only (results are garbage)

    const char* ta = smem + s * kStageBytes;  // rows of a   -> MFMA B operand (columns = m)
    const char* tb = ta + kTileBytes;         // rows of b^T -> MFMA A operand (rows = n)

    v8i nfr[4];
    if constexpr (MODE == MODE_BLOCKWISE && (VAR == 1 || VAR == 8)) {
      int fo = frag_off;
      asm volatile("" : "+v"(fo));
      const uint32_t b_lo = lds_base + (uint32_t)(s * kStageBytes + kTileBytes + wn * 64 * 128) + (uint32_t)fo;
      const uint32_t b_hi = b_lo ^ 64u;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) nfr[nf] = read_frag_lds(b_lo, b_hi, nf * 16 * 128);
    } else {
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) nfr[nf] = read_frag(tb, (wn * 64 + nf * 16) * 128 + frag_off);
    }

    if constexpr (MODE == MODE_BLOCKWISE && (VAR == 1 || VAR == 8)) {
      // V1: m-fragments prefetched one step ahead, next block's DMA issued in four parts between the MFMA clusters.
      // All LDS addresses of the block derive from ONE per-lane constant re-materialised here (opaque to LICM).
      int fo = frag_off;
      asm volatile("" : "+v"(fo));
      const uint32_t sbase = lds_base + (uint32_t)(s * kStageBytes);
      const uint32_t a_lo = sbase + (uint32_t)(wm * 128 * 128) + (uint32_t)fo, a_hi = a_lo ^ 64u;
      const uint32_t ts_addr = sbase + 2 * kTileBytes + (uint32_t)(wm * 128 * 4) + (uint32_t)((fo >> 7) << 2);
      typedef __attribute__((address_space(3))) const float* lds_f_ptr;
      // this block's column-block scale was fetched (scalar load) one block ago; pin it into a VGPR here, at the
      // top of the block, so that the lgkmcnt(0) a scalar-load consumer needs is paid before any LDS read is in flight
      float sbv = sbv_next;
      asm volatile("" : "+v"(sbv));
      if (kb + 1 < nkb) sbv_next = sb_wave[(int64_t)(kb + 1) * sb_sk];
      const v4f zero = {0.f, 0.f, 0.f, 0.f};
      constexpr int PD = 1;  // LDS prefetch distance in m-steps (2 measured no faster)
      constexpr int NB = PD + 1;
      v8i mfr[NB];
      float raw[NB];  // row scales are multiplied by sb at use, so the prefetch is not waited for early
#pragma unroll
      for (int p = 0; p < PD; ++p) {
        mfr[p] = read_frag_lds(a_lo, a_hi, p * 16 * 128);
        raw[p] = *(lds_f_ptr)(uintptr_t)(ts_addr + p * 64);
      }
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) {
        if (mf + PD < 8) {
          mfr[(mf + PD) % NB] = read_frag_lds(a_lo, a_hi, (mf + PD) * 16 * 128);
          raw[(mf + PD) % NB] = *(lds_f_ptr)(uintptr_t)(ts_addr + (mf + PD) * 64);
        }
        if (VAR == 1 && mf < 4 && kb + 1 < nkb) stage_part(kb + 1, s ^ 1, mf);
        const float sc = raw[mf % NB] * sbv;
        v4f cur[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) cur[nf] = mfma_k128<HW_SCALE>(nfr[nf], mfr[mf % NB], zero);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mf][nf][r] = __builtin_fmaf(cur[nf][r], sc, acc[mf][nf][r]);
        // keep the scheduler from pulling the LDS reads of later steps up here (register pressure -> spills)
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (MODE == MODE_BLOCKWISE) {
      const float* ts = reinterpret_cast<const float*>(ta + 2 * kTileBytes);
      const float sbv = sb_wave[(int64_t)kb * sb_sk];
      const v4f zero = {0.f, 0.f, 0.f, 0.f};
      v4f prev[4];
      float sprev = 0.f;
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) {
        const v8i mfr = read_frag(ta, (wm * 128 + mf * 16) * 128 + frag_off);
        const float sc = ts[wm * 128 + mf * 16 + j] * sbv;
        v4f cur[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) cur[nf] = mfma_k128<HW_SCALE>(nfr[nf], mfr, zero);
        if (mf > 0) {
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[mf - 1][nf][r] = __builtin_fmaf(prev[nf][r], sprev, acc[mf - 1][nf][r]);
        }
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) prev[nf] = cur[nf];
        sprev = sc;
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[7][nf][r] = __builtin_fmaf(prev[nf][r], sprev, acc[7][nf][r]);
    } else {
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) {
        const v8i mfr = read_frag(ta, (wm * 128 + mf * 16) * 128 + frag_off);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          if constexpr (MODE == MODE_INT8_ROWCOL) {
            acc[mf][nf] = mfma_i8_k128(nfr[nf], mfr, acc[mf][nf]);
          } else {
            acc[mf][nf] = mfma_k128<HW_SCALE>(nfr[nf], mfr, acc[mf][nf]);
          }
        }
      }
    }
  }

  // ---- epilogue: lane owns out[m = .. + j][n = .. + 4g .. 4g+3]
#pragma unroll
  for (int mf = 0; mf < 8; ++mf) {
    const int m = m0 + wm * 128 + mf * 16 + j;
    if (m >= M) continue;
    OutT* orow = out + (int64_t)m * ldc;
    float sam = 1.f;
    if constexpr (MODE != MODE_BLOCKWISE) sam = sa[m];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int n = n0 + wn * 64 + nf * 16 + g * 4;
      OutT v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (MODE == MODE_BLOCKWISE) {
          v[r] = (OutT)acc[mf][nf][r];
        } else {
          const int nn = (n + r < N) ? (n + r) : (N - 1);
          const float t = ((float)acc[mf][nf][r] * sam) * sb[nn];
          if constexpr (MODE == MODE_FP8_ROWCOL) {
            v[r] = (OutT)t;
            if (bias != nullptr) v[r] = (OutT)((float)v[r] + (float)bias[nn]);
          } else {
            v[r] = (bias != nullptr) ? (OutT)(t + (float)bias[nn]) : (OutT)t;
          }
        }
      }
      if constexpr (VEC_STORE) {
        if (n < N) {
          Vec<OutT, 4> vv;
#pragma unroll
          for (int r = 0; r < 4; ++r) vv[r] = v[r];
          store_vec<OutT, 4>(orow + n, vv);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) orow[n + r] = v[r];
      }
    }
  }
}

// buffer resource (raw, 32-bit records) from a wave-uniform pointer and byte count; the readfirstlanes are free when
// the compiler already knows the values are uniform and keep it from building a waterfall loop when it does not
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t nrec) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(nrec), 0x00020000);
}

// ---------------------------------------------------------------------------------------------------------
// Blockwise kernel, persistent and pipelined across K blocks AND tiles. Same tile, wave layout and arithmetic as
// above. One workgroup per CU walks its tiles (XCD-aware order); the sequence of K blocks of all its tiles is ONE
// software pipeline:
//  * the one barrier per K block sits between m-steps 6 and 7: by then every wave has read all it needs from the
//    running stage (the last m-fragment is in registers) and the DMA of the next block, issued in pieces during
//    m-steps 0..2 and the previous block's step 7, has had >1000 cycles to land;
//  * step 7 overlaps its four MFMAs with the LDS reads of the NEXT block's n-fragments (each into the registers
//    of the n-fragment the MFMA just issued has consumed) and first m-fragment, so the matrix pipe does not drain
//    at the block boundary; the first DMA pieces of block +2 go out in the same step;
//  * "next block" crosses tile boundaries: the first block of the next tile is in LDS / in registers before the
//    running tile ends;
//  * the epilogue of a tile is folded into the FIRST K block of the next one: m-step s of that block converts and
//    stores accumulator rows 2s, 2s+1 of the finished tile (s = 0..3) just before it overwrites them with
//    acc = partial * scale (no zeroing), so the 128 KiB of output per tile drain under MFMAs instead of stalling
//    all CUs at once.
// Output ownership: the n-fragment nf of a wave is made of LDS rows (nf>>1)*32 + (nf&1)*4 + (i>>2)*8 + (i&3),
// i = MFMA row. A lane (m row j, group g) then owns columns g*8..g*8+7 and 32+g*8..32+g*8+7 of its wave's 64:
// two 16-byte stores per m-fragment, 64 contiguous bytes per row per instruction. The b tile's LDS swizzle key is
// ((row>>3)&3)<<1 | (row>>1)&1 so that those 16 rows are conflict-free for ds_read_b128.
// LDS-DMA and the stores go through buffer resources: the per-lane address part is a loop-invariant VGPR,
// everything that changes (tile, K block, piece) is scalar; rows past the end of a / b / out are out of the
// resource's range (loads give zeros, stores are dropped); "nothing to do" is a resource with zero records.
// Every K-block body is ONE basic block (selects, no branches): with branches the optimiser sinks MFMAs across
// the hand-placed LDS reads and spills.
// Requires K >= 256, N % 8 == 0, ldc % 8 == 0, out 16-byte aligned (the host uses the kernel above otherwise).
// PROBE: 0 = real kernel; timing probes with garbage results: 1 = no DMA inside the loop, 2 = no DMA and no
// stores, 3 = no stores, 4 = no stores and every DMA re-fetches K block 0 (L2 hits).
struct TileDesc {
  const uint8_t* pa;  // a + m0 * lda
  const uint8_t* pb;  // b + n0 * ldb
  const float* ps;    // sa + m0 * sa_sm
  const float* sbw;   // sb + (this wave's 128-column block) * sb_sn
  void* po;           // out + m0 * ldc + n0
  uint32_t nrec_a, nrec_b, nrec_s, nrec_o;  // bytes in range of the four resources (0: nothing)
  int ncols;                                // valid columns of the tile (<= BN)
  int wrows;                                // rows per wm half of the wave grid: 128 (256-row tile) or 64 (128-row half tile)
  int m0, n0;                               // first row / column (row / column scale modes: the epilogue's scales and bias)
  int rev;                                  // (diagnostic build, probe 16) 1: the tile walks its K blocks from the last to the first
};

// MODE_FP8_ROWCOL / MODE_INT8_ROWCOL (fp8_scaled_mm / int8_scaled_mm): the same pipeline without the block scales - the
// MFMAs of a tile chain into its accumulators (int8: two K = 64 MFMAs per fragment pair), no promotion FMAs, no scale
// DMA; the first K block of the next tile fetches the finished tile's row scales, column scales and bias at its top and
// applies them in the stores (the oracle's rounding order). Requires sb and bias 16-byte aligned as well.
// (a PHASE of a kernel: the units of one kind - MS = 8 whole tiles or MS = 4 half tiles - of this workgroup, prologue to final
//  stores and drain; smem = the workgroup's kStages * kStageBytes of LDS. The kernels below run one phase, or - round 5 - the whole
//  tiles and then the half tiles of the last partial round in ONE launch, as gemm_fp8bw_x32_kernel does since round 4.)
template <typename OutT, int MODE, bool HW_SCALE, int PROBE, int MS>  // MS m-steps per K block: 8 = 256-row tiles, 4 = 128-row half tiles
__device__ __forceinline__ void gemm_8bit_persist_phase(
    char* smem, OutT* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ sa, const float* __restrict__ sb, const OutT* __restrict__ bias, int M, int N, int K,
    int64_t lda, int64_t ldb, int64_t ldc, int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, int tiles_m,
    int tiles_n, int all_halves, uint32_t* __restrict__ stamps, const void* __restrict__ x0,
    const void* __restrict__ x1, int ksplit = 1, int64_t slab = 0) {
  constexpr bool kBW = MODE == MODE_BLOCKWISE;
  constexpr bool kW4 = MODE == MODE_W4A8_CHN || MODE == MODE_W4A8_GRP, kGrp = MODE == MODE_W4A8_GRP;
  // K slices (row / column scale modes, OutT = float as a 4-byte carrier; see launch()): unit = tile x slice, slice s multiplies K
  // blocks [s nkb, (s + 1) nkb) and stores its RAW accumulators (fp32, or int32 bits for int8) into slab s of `out`; the sum kernel
  // adds the slabs and applies scales and bias in the epilogue's order (int8: bit-identical to the unsplit result).
  constexpr bool kRaw = sizeof(OutT) == 4 && !kBW && !kW4;
  constexpr bool kI8 = MODE == MODE_INT8_ROWCOL || kW4;
  using AccT = typename std::conditional<kI8, v4i, v4f>::type;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nkb = kRaw ? K / BK / ksplit : K / BK;  // >= 2
  constexpr bool kDma = PROBE == 0 || PROBE >= 3, kStore = PROBE == 0 || PROBE == 1 || PROBE >= 5;
  // schedule experiments with correct results (diagnostic build): 6 = all LDS-DMA issued by waves 0..3 (the older wave of
  // each SIMD, which wins the issue arbitration and otherwise waits ~900 cycles per K block at the barrier for the
  // younger one), 7 = 6 + static priority 1 for waves 4..7
  constexpr bool kLeadDma = PROBE == 6 || PROBE == 7;
  if constexpr (PROBE == 7) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }
  // PROBE 5 (diagnostic build only): three s_memtime stamps per K block (compute done / own DMA landed / barrier
  // released) for K blocks 40..60 of the workgroup, kept in the lanes of one VGPR and written out at the end
  uint32_t stampv = 0;

  // ---- this workgroup's tiles: workgroups b, b+8, ... share an XCD; each XCD owns a contiguous run of tiles,
  // walked in groups of 4 m-tiles so that the 32 tiles in flight on an XCD share a and b panels in its L2
  const int nt = tiles_m * tiles_n * (kRaw ? ksplit : 1);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int q = nt >> 3, rem = nt & 7;
  const int run_first = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int run_len = q + (xcd < rem ? 1 : 0);
  const int nblk_max = (N + 127) / 128 - 1;

  // Units of this workgroup. When the XCD's run of tiles does not divide by its workgroups and the remainder fits
  // twice, the last partial round is cut into 128-row half tiles and handed to a second launch of this kernel with
  // MS = 4 (same code, 4 m-steps per K block), so that it costs ~0.6 of a round instead of a whole one: 896 tiles on
  // 256 CUs = 3 rounds (launch MS = 8) + 128 tiles -> 256 halves, one per workgroup (launch MS = 4).
  const int rounds = run_len / slots, left = run_len - rounds * slots;
  const bool split = left > 0 && 2 * left <= slots;
  constexpr bool kHalf = MS == 4;
  // all_halves (MS = 4 only): every tile of the problem is processed as two 128-row halves, half tile h of the XCD's
  // run = (tile h / 2, half h % 2). Up to 512 rows this gives 2x the workgroups of the 256-row tiling (a 128-row
  // problem at N = 14336 has 56 tiles) at 0.6x the time per K block.
  const int n_units = (kHalf && all_halves) ? (2 * run_len > slot ? (2 * run_len - slot + slots - 1) / slots : 0)
                      : kHalf             ? ((split && slot < 2 * left) ? 1 : 0)
                                          : (split ? rounds : rounds + (slot < left ? 1 : 0));
  if (n_units == 0) return;

  auto describe = [&](int unit) -> TileDesc {  // unit >= n_units: the null tile
    TileDesc d;
    bool live = unit < n_units;
    constexpr bool half = kHalf;
    const int ht = slot + unit * slots;  // half-tile index in the all_halves walk
    const int local = !live ? 0 : (half && all_halves) ? (ht >> 1) : half ? rounds * slots + (slot >> 1) : slot + unit * slots;
    const int lower = all_halves ? (ht & 1) : (slot & 1);
    const int tile_s = run_first + local;
    const int tile = kRaw ? tile_s / ksplit : tile_s;
    const int kb0 = kRaw ? (tile_s - tile * ksplit) * nkb : 0;  // the slice's first K block
    constexpr int GM = 4;
    const int group = tile / (GM * tiles_n);
    const int first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_group = tile - group * GM * tiles_n;
    const int tm = __builtin_amdgcn_readfirstlane(first_m + in_group % gsz);
    const int tn = __builtin_amdgcn_readfirstlane(in_group / gsz);
    const int trows = half ? BM / 2 : BM;
    const int m0 = tm * BM + ((half && lower) ? BM / 2 : 0), n0 = tn * BN;
    const int rows_a = (M - m0) < trows ? (M - m0) : trows, rows_b = (N - n0) < BN ? (N - n0) : BN;
    live = live && rows_a > 0;  // the lower half of an edge tile may be empty
    d.ncols = rows_b;
    d.wrows = MS * 16;
    d.m0 = m0;
    d.n0 = n0;
    d.pa = a + (int64_t)m0 * lda + kb0 * BK;
    d.pb = kW4 ? b + (int64_t)n0 * (K >> 1) : b + (int64_t)n0 * ldb + kb0 * BK;  // (W4: 32-column groups of K/32 512-byte blocks)
    d.ps = kBW ? sa + (int64_t)m0 * sa_sm : sa;
    d.po = (void*)(out + (kRaw ? (int64_t)(kb0 / nkb) * slab : 0) + (int64_t)m0 * ldc + n0);
    const int kdepth = kRaw ? nkb * BK : K;
    d.nrec_a = live ? (uint32_t)((int64_t)(rows_a - 1) * lda + kdepth) : 0u;
    d.nrec_b = !live ? 0u : kW4 ? (uint32_t)((int64_t)(rows_b >> 5) * (K >> 5) * 512) : (uint32_t)((int64_t)(rows_b - 1) * ldb + kdepth);
    // row scales: waves 0..3 fetch 64 rows each (4 B per lane); waves 4..7 fetch nothing (zeros into the spare KiB)
    d.nrec_s = (kBW && live && wave < 4) ? (uint32_t)(((int64_t)(rows_a - 1) * sa_sm + (int64_t)(nkb - 1) * sa_sk + 1) * 4) : 0u;
    d.nrec_o = (live && kStore) ? (uint32_t)(((int64_t)(rows_a - 1) * ldc + rows_b) * (int64_t)sizeof(OutT)) : 0u;
    int nblk = (n0 + wn * 64) >> 7;
    nblk = nblk < nblk_max ? nblk : nblk_max;
    d.sbw = kBW ? sb + (int64_t)nblk * sb_sn : sb;
    d.rev = 0;
    return d;
  };
  auto pick = [](bool c, const TileDesc& x, const TileDesc& y) -> TileDesc {  // scalar selects
    TileDesc d;
    d.pa = c ? x.pa : y.pa;  d.pb = c ? x.pb : y.pb;  d.ps = c ? x.ps : y.ps;  d.sbw = c ? x.sbw : y.sbw;
    d.po = c ? x.po : y.po;
    d.nrec_a = c ? x.nrec_a : y.nrec_a;  d.nrec_b = c ? x.nrec_b : y.nrec_b;  d.nrec_s = c ? x.nrec_s : y.nrec_s;
    d.nrec_o = c ? x.nrec_o : y.nrec_o;  d.ncols = c ? x.ncols : y.ncols;  d.wrows = c ? x.wrows : y.wrows;
    d.m0 = c ? x.m0 : y.m0;  d.n0 = c ? x.n0 : y.n0;
    d.rev = 0;
    return d;
  };

  const uint32_t lds_base = (uint32_t)(uintptr_t)SGLK_LDS(smem);
  // piece p of a tile covers rows 8p..8p+7; lane -> row 8p + lane/8, 16-byte chunk (lane%8) ^ key(row):
  //   a: key = (row>>1)&7            = (4(p&1) + lane/16) & 7      -> depends on the parity of p
  //   b: key = ((row>>3)&3)<<1 | (row>>1)&1 = (p&3)<<1 | (lane/16)&1 -> depends on p&3 (p = 4*wave + ii: on ii)
  uint32_t voff_a[2], voff_b[4];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    voff_a[par] = (uint32_t)(lane >> 3) * (uint32_t)lda + (((lane & 7) ^ ((par * 4 + (lane >> 4)) & 7)) << 4);
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
    voff_b[ii] = (uint32_t)(lane >> 3) * (uint32_t)ldb + (((lane & 7) ^ ((ii << 1) | ((lane >> 4) & 1))) << 4);
  const uint32_t voff_s = (uint32_t)tid * (uint32_t)sa_sm * 4u;
  // W4: a 1-KiB piece = two 512-byte blocks (k32 = 2 ks, 2 ks + 1 of one 32-column group); its 16-byte chunks (blk, c, e)
  // land at e * 256 + blk * 128 + c * 16, so that the fragment read of one e touches 256 contiguous bytes
  const uint32_t voff_w = (uint32_t)(((lane >> 3) & 1) * 512 + (lane & 7) * 64 + (lane >> 4) * 16);

  // LDS-DMA of K block kb of tile d into stage s, one 1-KiB piece per call: part 0, 1 = rows of a, part 2, 3 = rows
  // of b^T, two pieces (sub 0, 1) each; (part 0, sub 2) = the block's row scales
  auto dma_piece = [&](const TileDesc& d, int kb_, int s, int part, int sub) {
    const int kb = PROBE == 4 ? 0 : kb_;  // probe 4: every block re-fetches block 0 (L2 hits, no stores)
    char* base = smem + s * kStageBytes;
    if (sub == 2) {
      if (kBW && (!kLeadDma || wave < 4))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(d.ps, d.nrec_s),
                                                 SGLK_LDS(base + 2 * kTileBytes + wave * 256), 4, voff_s,
                                                 kb * (int)sa_sk * 4, 0, 0);
      return;
    }
    if constexpr (kW4) {
      if (part == 3) return;
      if (part == 2) {  // the wave's 32-column group: k steps sub = 0, 1 of the K block
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(d.pb, d.nrec_b),
                                                 SGLK_LDS(base + kTileBytes + (wave * 2 + sub) * 1024), 16, voff_w,
                                                 (wave * (K >> 5) + kb * 4 + sub * 2) * 512, 0, 0);
        return;
      }
    }
    const int ii = (part & 1) * 2 + sub;
    auto one = [&](int piece) {
      if (part < 2) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(d.pa, d.nrec_a), SGLK_LDS(base + piece * 1024), 16,
                                                 voff_a[ii & 1], kb * BK + piece * 8 * (int)lda, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(d.pb, d.nrec_b), SGLK_LDS(base + kTileBytes + piece * 1024),
                                                 16, voff_b[ii], kb * BK + piece * 8 * (int)ldb, 0, 0);
      }
    };
    if constexpr (kLeadDma) {
      if (wave < 4) {
        one(wave * 4 + ii);
        one((wave + 4) * 4 + ii);
      }
    } else {
      one(wave * 4 + ii);
    }
  };

  const int j = lane & 15, g = lane >> 4;
  const int frag_off_a = j * 128 + ((g ^ ((j >> 1) & 7)) << 4);
  const int frag_off_b = ((j >> 2) * 8 + (j & 3)) * 128 + ((g ^ ((j >> 1) & 7)) << 4);
  constexpr int kNfImm[4] = {0, 4 * 128, 32 * 128, 36 * 128};  // first LDS row of n-fragment nf, in bytes
  typedef __attribute__((address_space(3))) const float* lds_f_ptr;

  // stores: lane (j, g) owns row wm*128 + mf*16 + j, columns wn*64 + h*32 + g*8 .. +7 (h = 0, 1) of the tile
  const uint32_t orow_off_full = (uint32_t)(((int64_t)(wm * 128 + j) * ldc + wn * 64 + g * 8) * (int64_t)sizeof(OutT));
  const uint32_t orow_off_half = (uint32_t)(((int64_t)(wm * 64 + j) * ldc + wn * 64 + g * 8) * (int64_t)sizeof(OutT));
  // row / column scale modes: what the stores of a finished tile need (fetched at the top of the next tile's first K block)
  struct Epi {
    float sa[8];
    v4f sb[2][2];
    Vec<OutT, 8> bias[2];
    // W4: lane (j, g) owns row j of an m-fragment and columns wn*64 + (nf>>1)*32 + (nf&1)*16 + 4 g .. + 3 of n-fragment nf
    float asum[8];
    Vec<f16, 4> sw[4], wz[4];
  };
  auto load_epi = [&](const TileDesc& d) -> Epi {
    Epi e;
    if constexpr (kW4) {
      const f16* ascales = reinterpret_cast<const f16*>(sa);
      const f16* wscales = reinterpret_cast<const f16*>(sb);
#pragma unroll
      for (int mf = 0; mf < MS; ++mf) {
        int m = d.m0 + wm * (MS * 16) + mf * 16 + j;
        m = m < M ? m : M - 1;
        e.sa[mf] = (float)ascales[m];
        if constexpr (!kGrp) e.asum[mf] = (float)reinterpret_cast<const f16*>(x0)[m];
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        int n = d.n0 + wn * 64 + (nf >> 1) * 32 + (nf & 1) * 16 + g * 4;  // (N % 32 == 0: in range or dropped as a whole)
        n = n < N ? n : 0;
        e.sw[nf] = load_vec<f16, 4>(wscales + n);
        if constexpr (!kGrp) e.wz[nf] = load_vec<f16, 4>(reinterpret_cast<const f16*>(bias) + n);
      }
      return e;
    }
#pragma unroll
    for (int mf = 0; mf < MS; ++mf) {
      int m = d.m0 + wm * (MS * 16) + mf * 16 + j;
      m = m < M ? m : M - 1;
      e.sa[mf] = sa[m];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int n = d.n0 + wn * 64 + h * 32 + g * 8;  // (N % 8 == 0: the 8 columns are all in range or all dropped by the store)
      n = n < N ? n : 0;
      e.sb[h][0] = *reinterpret_cast<const v4f*>(sb + n);
      e.sb[h][1] = *reinterpret_cast<const v4f*>(sb + n + 4);
      if (bias != nullptr) {
        e.bias[h] = load_vec<OutT, 8>(bias + n);
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) e.bias[h][c] = (OutT)0.f;
      }
    }
    return e;
  };
  auto store_rows = [&](const TileDesc& d, const auto& accm, int mf, const Epi& e) {
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(d.po, d.nrec_o);
    const int soff = __builtin_amdgcn_readfirstlane(mf * 16 * (int)ldc * (int)sizeof(OutT));
    const uint32_t orow_off = MS == 8 ? orow_off_full : orow_off_half;
    if constexpr (kW4) {
      // (orow_off addresses column wn*64 + 8 g; this layout wants wn*64 + 4 g + 16 (nf&1) + 32 (nf>>1))
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        Vec<f16, 4> v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = (float)accm[nf][r] * e.sa[mf] * (float)e.sw[nf][r];  // (the small kernels' / the oracle's order)
          if constexpr (!kGrp) t -= e.asum[mf] * (float)e.wz[nf][r];
          v[r] = (f16)t;
        }
        const int col = wn * 64 + (nf >> 1) * 32 + (nf & 1) * 16 + g * 4;
        const uint32_t vo = col < d.ncols ? orow_off + (uint32_t)(((nf >> 1) * 32 + (nf & 1) * 16 - g * 4) * 2) : 0x80000000u;
        const v2i data = __builtin_bit_cast(v2i, v);
        __builtin_amdgcn_raw_buffer_store_b64(data, ro, (int)vo, soff, 0);
        asm volatile("s_nop 4" ::"v"(data));  // (see below)
      }
      return;
    }
    if constexpr (kRaw) {  // the lane's 8 accumulators of a fragment pair as they are: two 16-byte stores
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t vo = (wn * 64 + h * 32 + g * 8 < d.ncols) ? orow_off + h * 32 * 4u : 0x80000000u;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const auto a4 = accm[2 * h + q2];
          const v4i data = __builtin_bit_cast(v4i, a4);
          __builtin_amdgcn_raw_buffer_store_b128(data, ro, (int)vo + q2 * 16, soff, 0);
          asm volatile("s_nop 4" ::"v"(data));
        }
      }
    } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      Vec<OutT, 8> v;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if constexpr (kBW) {
          v[c] = (OutT)accm[2 * h + (c >> 2)][c & 3];
        } else {
          // (the tile kernel's / the oracle's rounding order)
          const float t = ((float)accm[2 * h + (c >> 2)][c & 3] * e.sa[mf]) * e.sb[h][c >> 2][c & 3];
          if constexpr (MODE == MODE_FP8_ROWCOL) {
            v[c] = (OutT)t;
            if (bias != nullptr) v[c] = (OutT)((float)v[c] + (float)e.bias[h][c]);
          } else {
            v[c] = (bias != nullptr) ? (OutT)(t + (float)e.bias[h][c]) : (OutT)t;
          }
        }
      }
      // a column past the tile's valid ones would land in the next row: push those lanes out of range instead
      const uint32_t vo = (wn * 64 + h * 32 + g * 8 < d.ncols) ? orow_off + h * 32 * (uint32_t)sizeof(OutT) : 0x80000000u;
      const v4i data = __builtin_bit_cast(v4i, v);
      __builtin_amdgcn_raw_buffer_store_b128(data, ro, (int)vo, soff, 0);
      // gfx950 reads the 16 bytes of store data for a few cycles after issue; the compiler assumes the form with
      // a scalar offset register has no such hazard and lets the next VALU op overwrite the registers (seen:
      // dword 1 of lanes 12-15 of each row corrupted). Keep them alive across a short nop.
      asm volatile("s_nop 4" ::"v"(data));
    }
    }
  };

  // Accumulators as scalars (blockwise: the promotion FMAs below are inline asm on single registers) or as the MFMAs'
  // C / D operands (row / column scale modes).
  typename std::conditional<kBW, float[8][4][4], AccT[8][4]>::type acc;
#pragma unroll
  for (int mf = 0; mf < 8; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mf][nf][r] = 0;
  Epi epi;

  // The main loop is written as a fixed instruction stream: LDS reads, MFMAs and the block-scale FMAs are inline asm
  // (asm volatile statements keep their order; the compiler only allocates registers), the waits are counted by hand
  // and every wait names the registers it releases so that their consumers stay below it.
  //  * an m-step is four slots "MFMA; 4 x v_fma (promotion of the PREVIOUS m-step's partial of the same n-fragment);
  //    one LDS read or LDS-DMA piece": no FMA waits on an MFMA issued less than an m-step (>= 4 MFMAs) earlier, so an
  //    in-order wave never idles on MFMA latency and the compiler's MFMA -> VALU hazard nops have nothing to cover.
  //    (The first version issued 4 MFMAs and then their 16 dependent FMAs: both waves of a SIMD stalled on their own
  //    results at the same time and the matrix pipe was busy 53 % of the block; measured per wave with s_memtime:
  //    2700 / 3450 cycles per K block for the older / younger wave of a SIMD against 2048 of MFMA time.)
  //  * the promotion runs one m-step behind across K blocks and tiles: step 0 of a block promotes the last m-step of the
  //    block before (cur[] and scp carry it), the kernel's tail promotes the very last one.
  //  * the first K block of a tile stores the finished tile's rows 2s, 2s+1 in m-step s and zeroes them; row MS-1 of
  //    the finished tile is completed by that block's step 0 and stored in step (MS-1)/2.
#define SGLK_RD16(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define SGLK_RD4(dst, addr, imm) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define SGLK_FRAG(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)
// cur[cb][nf] = n-fragment nf x m-fragment buffer mb (zero C: the partial of ONE 128-deep block)
#define SGLK_MFMA(cb, nf, mb, row)                                                                             \
  if constexpr (kI8) {                                                                    \
    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0\n\tv_mfma_i32_16x16x64_i8 %0, %3, %4, %0"               \
                 : "+v"(acc[row][nf])                                                                          \
                 : "v"(nlo[nf]), "v"(mlo[mb]), "v"(nhi[nf]), "v"(mhi[mb]));                                    \
  } else if constexpr (MODE == MODE_FP8_ROWCOL && HW_SCALE) {                                                  \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"                  \
                 : "+v"(acc[row][nf])                                                                          \
                 : "v"(SGLK_FRAG(nlo[nf], nhi[nf])), "v"(SGLK_FRAG(mlo[mb], mhi[mb])), "v"(one_e8m0));         \
  } else if constexpr (MODE == MODE_FP8_ROWCOL) {                                                              \
    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0"                                                  \
                 : "+v"(acc[row][nf])                                                                          \
                 : "v"(SGLK_FRAG(nlo[nf], nhi[nf])), "v"(SGLK_FRAG(mlo[mb], mhi[mb])));                        \
  } else if constexpr (HW_SCALE)                                                                               \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"                   \
                 : "=&v"(cur[cb][nf])                                                                          \
                 : "v"(SGLK_FRAG(nlo[nf], nhi[nf])), "v"(SGLK_FRAG(mlo[mb], mhi[mb])), "v"(one_e8m0));         \
  else                                                                                                         \
    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0"                                                   \
                 : "=&v"(cur[cb][nf])                                                                          \
                 : "v"(SGLK_FRAG(nlo[nf], nhi[nf])), "v"(SGLK_FRAG(mlo[mb], mhi[mb])));
// acc[row][nf][:] += cur[cb][nf][:] * scp
#define SGLK_PROMOTE(row, cb, nf)                                                                              \
  if constexpr (kBW) {                                                                                         \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                           \
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[row][nf][r_]) : "v"(cur[cb][nf][r_]), "v"(scp));    \
  }

  v4i nlo[4], nhi[4], mlo[2], mhi[2];
  float raw[2];
  // ---- W4 (QServe): the packed 4-bit weights of the K block sit in LDS as they are in memory (the 32 x 32 block IS an MFMA
  // layout: dword e of lane (j = 8 b + c, kg) = k 16 kg + 4 e .. + 3 of row j in the low and of row 16 + j in the high
  // nibbles). rw[q][ks][p] = dwords e = 2 p, 2 p + 1 of the wave's 32-column group q, 64-deep k step ks; one AND and one
  // shift + AND per dword make n-fragments 2 q (rows j) and 2 q + 1 (rows 16 + j): nlo = step 0, nhi = step 1.
  // Per group: the K block is the group; (code * s8 + zs8) is formed as u8 + 128 in ONE packed 16-bit multiply-add per
  // dword (code * s8 + (zs8 + 128) is in 1..255: no carries between bytes) and flipped back to int8 by an XOR.
  v2i rw[2][2][2];
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  uint32_t smul[2][2] = {{0, 0}, {0, 0}}, zadd[2][2] = {{0, 0}, {0, 0}};  // [q][half]: s8 in both 16-bit halves, zs8 + 128 in all bytes
  uint32_t szraw[2][2] = {{0, 0}, {0, 0}};                                 // [q][s8 / zs8]: the dword of columns c, 8+c, 16+c, 24+c
  const int wj_c = j & 7, wj_b = j >> 3;
  const uint32_t w_lane = (uint32_t)((g >> 1) * 128 + wj_c * 16 + (g & 1) * 8 + wj_b * 4 + wn * 4096);
  auto load_sz = [&](const TileDesc& d, int kb) {  // group scales / zero terms of K block kb -> szraw (used one block later)
    if constexpr (kGrp) {
      const __amdgpu_buffer_rsrc_t r0 = make_rsrc(x0, (uint32_t)((int64_t)(K >> 7) * N));
      const __amdgpu_buffer_rsrc_t r1 = make_rsrc(x1, (uint32_t)((int64_t)(K >> 7) * N));
      const int so = __builtin_amdgcn_readfirstlane(kb * N + d.n0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        szraw[q][0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r0, wn * 64 + q * 32 + wj_c * 4, so, 0);
        szraw[q][1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r1, wn * 64 + q * 32 + wj_c * 4, so, 0);
      }
    }
  };
  auto prep_sz = [&]() {
    if constexpr (kGrp) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t sv = (szraw[q][0] >> ((2 * h + wj_b) * 8)) & 0xffu;
          const uint32_t zv = ((szraw[q][1] >> ((2 * h + wj_b) * 8)) & 0xffu) ^ 0x80u;
          smul[q][h] = sv | (sv << 16);
          const uint32_t z2 = zv | (zv << 8);
          zadd[q][h] = z2 | (z2 << 16);
        }
    }
  };
  auto unpack_w = [&](const v2i (&r)[2], v4i& flo, v4i& fhi, int q) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t wd = (uint32_t)r[e >> 1][e & 1];
      uint32_t lo = wd & 0x0f0f0f0fu, hi = (wd >> 4) & 0x0f0f0f0fu;
      if constexpr (kGrp) {
        lo = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, lo) * __builtin_bit_cast(u16x2, smul[q][0]) +
                                                  __builtin_bit_cast(u16x2, zadd[q][0]))) ^ 0x80808080u;
        hi = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, hi) * __builtin_bit_cast(u16x2, smul[q][1]) +
                                                  __builtin_bit_cast(u16x2, zadd[q][1]))) ^ 0x80808080u;
      }
      flo[e] = (int)lo;
      fhi[e] = (int)hi;
    }
  };
// the eight fragment reads of a K block: ds_read2st64_b32 (offsets in units of 256 B: piece (q, ks) at 2 q + ks KiB, e at 256 e)
#define SGLK_RDW(q, ks, addr)                                                                                  \
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(rw[q][ks][0]) : "v"(addr), "n"((q) * 8 + (ks) * 4), "n"((q) * 8 + (ks) * 4 + 1)); \
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(rw[q][ks][1]) : "v"(addr), "n"((q) * 8 + (ks) * 4 + 2), "n"((q) * 8 + (ks) * 4 + 3));
  float sbv;
  v4f cur[2][4];    // partials of the running (mf & 1) and the previous m-step
  float scp = 0.f;  // row scale x column-block scale of the previous m-step
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) cur[1][nf] = (v4f){0.f, 0.f, 0.f, 0.f};  // (the first step 0 promotes nothing)
  int one_e8m0 = 127;  // E8M0 2^0 for both operands of the MX-encoded MFMA
  asm volatile("" : "+v"(one_e8m0), "+v"(scp), "+v"(cur[1][0]), "+v"(cur[1][1]), "+v"(cur[1][2]), "+v"(cur[1][3]));
  int gblk = 0;  // K blocks done so far by this workgroup: its parity is the running LDS stage

  // m-step mf < MS - 1 of a K block (see above). Reads for m-fragment mf + 1 go out first (the buffer they land in was
  // last read by the MFMAs of step mf - 1, all issued), then fragment mf is waited for.
#define SGLK_STEP(mf, STORE, MS)                                                                               \
  if constexpr ((mf) < (MS) - 1) {                                                                             \
    constexpr int cb_ = (mf) & 1, pb_ = cb_ ^ 1, prow_ = ((mf) + (MS) - 1) % (MS);                             \
    SGLK_RD16(mlo[pb_], a_lo, ((mf) + 1) * 2048);                                                              \
    SGLK_RD16(mhi[pb_], a_hi, ((mf) + 1) * 2048);                                                              \
    SGLK_RD4(raw[pb_], ts_addr, ((mf) + 1) * 64);                                                              \
    if ((mf) == 0) {                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(nlo[0]), "+v"(nhi[0]), "+v"(mlo[0]), "+v"(mhi[0]), "+v"(raw[0])); \
    } else {                                                                                                   \
      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(mlo[cb_]), "+v"(mhi[cb_]), "+v"(raw[cb_]));                   \
    }                                                                                                          \
    float sc_ = 0.f;                                                                                           \
    if constexpr (kBW) sc_ = raw[cb_] * sbv;                                                                   \
    if constexpr (STORE && 2 * (mf) + 1 < (MS)) {                                                              \
      constexpr int r0_ = 2 * (mf) + 1 < (MS) ? 2 * (mf) : 0, r1_ = r0_ + 1; /* (in range also when discarded) */ \
      store_rows(prv, acc[r0_], r0_, epi);                                                                     \
      store_rows(prv, acc[r1_], r1_, epi);                                                                     \
      _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) _Pragma("unroll") for (int r = 0; r < 4; ++r) {         \
        acc[r0_][nf][r] = 0;                                                                                   \
        acc[r1_][nf][r] = 0;                                                                                   \
      }                                                                                                        \
      if constexpr (kBW) {                                                                                     \
        asm volatile("" : "+v"(acc[r0_][0][0]), "+v"(acc[r1_][0][0]));                                         \
      } else {                                                                                                 \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) asm volatile("" : "+v"(acc[r0_][nf]), "+v"(acc[r1_][nf])); \
      }                                                                                                        \
    }                                                                                                          \
    SGLK_MFMA(cb_, 0, cb_, (mf))                                                                               \
    SGLK_PROMOTE(prow_, pb_, 0)                                                                                \
    if constexpr (kW4 && (mf) == 0) { /* the last quarter of the block's weight unpack (see the last m-step) */ \
      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(rw[1][1][0]), "+v"(rw[1][1][1]));                             \
      unpack_w(rw[1][1], nhi[2], nhi[3], 1);                                                                   \
      asm volatile("" : "+v"(nhi[2]), "+v"(nhi[3]));                                                           \
    }                                                                                                          \
    if ((mf) == 0) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(nlo[1]), "+v"(nhi[1]));                          \
    SGLK_MFMA(cb_, 1, cb_, (mf))                                                                               \
    SGLK_PROMOTE(prow_, pb_, 1)                                                                                \
    if (kDma && (mf) < 3) dma_piece(d1, kb1, s ^ 1, (mf) + 1, 0);                                              \
    if ((mf) == 0) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(nlo[2]), "+v"(nhi[2]));                          \
    SGLK_MFMA(cb_, 2, cb_, (mf))                                                                               \
    SGLK_PROMOTE(prow_, pb_, 2)                                                                                \
    if ((mf) == 0) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(nlo[3]), "+v"(nhi[3]));                          \
    SGLK_MFMA(cb_, 3, cb_, (mf))                                                                               \
    SGLK_PROMOTE(prow_, pb_, 3)                                                                                \
    if (kDma && (mf) < 3) dma_piece(d1, kb1, s ^ 1, (mf) + 1, 1);                                              \
    if constexpr (kBW) {                                                                                       \
      scp = sc_;                                                                                               \
      asm volatile("" : "+v"(scp));                                                                            \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
#define SGLK_BLOCK(STORE, MS)                                                                                  \
  {                                                                                                            \
    const int s = gblk & 1;                                                                                    \
    const uint32_t sbase = lds_base + (uint32_t)(s * kStageBytes);                                             \
    const uint32_t nbase = lds_base + (uint32_t)((s ^ 1) * kStageBytes);                                       \
    const bool in1 = kb + 1 < nkb, in2 = kb + 2 < nkb;                                                         \
    const TileDesc d1 = pick(in1, cur_t, nxt), d2 = pick(in2, cur_t, nxt);                                     \
    const int kb1 = in1 ? kb + 1 : 0, kb2 = in2 ? kb + 2 : kb + 2 - nkb;                                       \
    if constexpr (STORE && !kBW) epi = load_epi(prv);                                                          \
    load_sz(d1, kb1);                                                                                          \
    {                                                                                                          \
      int fo = frag_off_a;                                                                                     \
      asm volatile("" : "+v"(fo));                                                                             \
      const uint32_t a_lo = sbase + (uint32_t)(wm * ((MS) * 16) * 128) + (uint32_t)fo, a_hi = a_lo ^ 64u;       \
      const uint32_t ts_addr = sbase + 2 * kTileBytes + (uint32_t)(wm * ((MS) * 16) * 4) + (uint32_t)((fo >> 7) << 2); \
      SGLK_STEP(0, STORE, MS) SGLK_STEP(1, STORE, MS) SGLK_STEP(2, STORE, MS) SGLK_STEP(3, STORE, MS)          \
      SGLK_STEP(4, STORE, MS) SGLK_STEP(5, STORE, MS) SGLK_STEP(6, STORE, MS)                                  \
    }                                                                                                          \
    /* the block's barrier: next block landed everywhere, nobody reads stage s any more */                    \
    float sbv_next = 0.f;                                                                                      \
    if constexpr (kBW) sbv_next = d1.sbw[(int64_t)kb1 * sb_sk];                                                \
    if constexpr (PROBE == 5) {                                                                                \
      const uint32_t t0 = (uint32_t)__builtin_amdgcn_s_memtime();                                              \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(mlo[1]), "+v"(mhi[1]), "+v"(raw[1]) : : "memory");   \
      const uint32_t t1 = (uint32_t)__builtin_amdgcn_s_memtime();                                              \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
      const uint32_t t2 = (uint32_t)__builtin_amdgcn_s_memtime();                                              \
      const int w_ = gblk - 40;                                                                                \
      const int sl_ = (w_ >= 0 && w_ < 20) ? w_ * 3 : 61;                                                      \
      stampv = (lane == sl_) ? t0 : stampv;                                                                    \
      stampv = (lane == sl_ + 1) ? t1 : stampv;                                                                \
      stampv = (lane == sl_ + 2) ? t2 : stampv;                                                                \
    } else {                                                                                                   \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"                                                \
                   : "+v"(mlo[1]), "+v"(mhi[1]), "+v"(raw[1])                                                  \
                   :                                                                                           \
                   : "memory");                                                                                \
    }                                                                                                          \
    asm volatile("" : "+v"(sbv_next));                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    /* last m-step: its slots carry the first LDS reads of the next block (each n-fragment into the registers */ \
    /* the MFMA just issued has consumed) and the first DMA pieces of the block after it                      */ \
    {                                                                                                          \
      constexpr int kLast = (MS) - 1;                                                                          \
      float sc_ = 0.f;                                                                                         \
      if constexpr (kBW) sc_ = raw[1] * sbv;                                                                   \
      int foa = frag_off_a, fob = frag_off_b;                                                                  \
      asm volatile("" : "+v"(foa), "+v"(fob));                                                                 \
      uint32_t nb_lo = nbase + (uint32_t)(kTileBytes + wn * 64 * 128) + (uint32_t)fob, nb_hi = nb_lo ^ 64u;   \
      uint32_t na_lo = nbase + (uint32_t)(wm * d1.wrows * 128) + (uint32_t)foa, na_hi = na_lo ^ 64u;           \
      uint32_t nts = nbase + 2 * kTileBytes + (uint32_t)(wm * d1.wrows * 4) + (uint32_t)((foa >> 7) << 2);    \
      asm volatile("" : "+v"(nb_lo), "+v"(nb_hi), "+v"(na_lo), "+v"(na_hi), "+v"(nts));                        \
      if constexpr (kW4) {                                                                                     \
        uint32_t nw = nbase + (uint32_t)kTileBytes + w_lane;                                                   \
        asm volatile("" : "+v"(nw));                                                                           \
        prep_sz();                                                                                             \
        SGLK_MFMA(1, 0, 1, kLast)                                                                              \
        SGLK_RDW(0, 0, nw) SGLK_RDW(0, 1, nw)                                                                  \
        SGLK_RD16(mlo[0], na_lo, 0);          SGLK_RD16(mhi[0], na_hi, 0);                                     \
        SGLK_MFMA(1, 1, 1, kLast)                                                                              \
        SGLK_RDW(1, 0, nw) SGLK_RDW(1, 1, nw)                                                                  \
        if (kDma) dma_piece(d2, kb2, s, 0, 0);                                                                 \
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(rw[0][0][0]), "+v"(rw[0][0][1]));                           \
        unpack_w(rw[0][0], nlo[0], nlo[1], 0);                                                                 \
        asm volatile("" : "+v"(nlo[0]), "+v"(nlo[1]));                                                         \
        SGLK_MFMA(1, 2, 1, kLast)                                                                              \
        if (kDma) dma_piece(d2, kb2, s, 0, 1);                                                                 \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(rw[0][1][0]), "+v"(rw[0][1][1]));                           \
        unpack_w(rw[0][1], nhi[0], nhi[1], 0);                                                                 \
        asm volatile("" : "+v"(nhi[0]), "+v"(nhi[1]));                                                         \
        SGLK_MFMA(1, 3, 1, kLast)                                                                              \
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(rw[1][0][0]), "+v"(rw[1][0][1]), "+v"(mlo[0]), "+v"(mhi[0])); \
        unpack_w(rw[1][0], nlo[2], nlo[3], 1);                                                                 \
        asm volatile("" : "+v"(nlo[2]), "+v"(nlo[3]));                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
      } else {                                                                                                 \
      SGLK_MFMA(1, 0, 1, kLast)                                                                                \
      SGLK_PROMOTE(kLast - 1, 0, 0)                                                                            \
      SGLK_RD16(nlo[0], nb_lo, kNfImm[0]);  SGLK_RD16(nhi[0], nb_hi, kNfImm[0]);                               \
      SGLK_RD16(mlo[0], na_lo, 0);          SGLK_RD16(mhi[0], na_hi, 0);                                       \
      SGLK_RD4(raw[0], nts, 0);                                                                                \
      SGLK_MFMA(1, 1, 1, kLast)                                                                                \
      SGLK_PROMOTE(kLast - 1, 0, 1)                                                                            \
      SGLK_RD16(nlo[1], nb_lo, kNfImm[1]);  SGLK_RD16(nhi[1], nb_hi, kNfImm[1]);                               \
      if (kDma) dma_piece(d2, kb2, s, 0, 0);                                                                   \
      SGLK_MFMA(1, 2, 1, kLast)                                                                                \
      SGLK_PROMOTE(kLast - 1, 0, 2)                                                                            \
      SGLK_RD16(nlo[2], nb_lo, kNfImm[2]);  SGLK_RD16(nhi[2], nb_hi, kNfImm[2]);                               \
      if (kDma) dma_piece(d2, kb2, s, 0, 1);                                                                   \
      SGLK_MFMA(1, 3, 1, kLast)                                                                                \
      SGLK_PROMOTE(kLast - 1, 0, 3)                                                                            \
      SGLK_RD16(nlo[3], nb_lo, kNfImm[3]);  SGLK_RD16(nhi[3], nb_hi, kNfImm[3]);                               \
      if (kDma) dma_piece(d2, kb2, s, 0, 2);                                                                   \
      if constexpr (kBW) {                                                                                     \
        scp = sc_;                                                                                             \
        sbv = sbv_next;                                                                                        \
        asm volatile("" : "+v"(scp));                                                                          \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      }                                                                                                        \
    }                                                                                                          \
    ++gblk;                                                                                                    \
  }

  int unit = 0;
  TileDesc cur_t = describe(0);
  TileDesc prv = describe(n_units);  // the null tile: nothing to store yet
  // ---- prologue: block 0 of the first unit lands, its resident fragments are read, part 0 of block 1 goes out
#pragma unroll
  for (int part = 0; part < 4; ++part) {
    dma_piece(cur_t, 0, 0, part, 0);
    dma_piece(cur_t, 0, 0, part, 1);
  }
  dma_piece(cur_t, 0, 0, 0, 2);
  sbv = kBW ? cur_t.sbw[0] : 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  asm volatile("" : "+v"(sbv));
  {
    const uint32_t b_lo = lds_base + (uint32_t)(kTileBytes + wn * 64 * 128) + (uint32_t)frag_off_b, b_hi = b_lo ^ 64u;
    const uint32_t a_lo = lds_base + (uint32_t)(wm * cur_t.wrows * 128) + (uint32_t)frag_off_a, a_hi = a_lo ^ 64u;
    const uint32_t ts0 = lds_base + 2 * kTileBytes + (uint32_t)(wm * cur_t.wrows * 4) + (uint32_t)(j << 2);
    if constexpr (kW4) {
      const uint32_t w0 = lds_base + (uint32_t)kTileBytes + w_lane;
      load_sz(cur_t, 0);
      SGLK_RDW(0, 0, w0) SGLK_RDW(0, 1, w0) SGLK_RDW(1, 0, w0) SGLK_RDW(1, 1, w0)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                   : "+v"(rw[0][0][0]), "+v"(rw[0][0][1]), "+v"(rw[0][1][0]), "+v"(rw[0][1][1]), "+v"(rw[1][0][0]),
                     "+v"(rw[1][0][1]), "+v"(rw[1][1][0]), "+v"(rw[1][1][1]), "+v"(szraw[0][0]), "+v"(szraw[0][1]),
                     "+v"(szraw[1][0]), "+v"(szraw[1][1]));
      prep_sz();
      unpack_w(rw[0][0], nlo[0], nlo[1], 0);
      unpack_w(rw[0][1], nhi[0], nhi[1], 0);
      unpack_w(rw[1][0], nlo[2], nlo[3], 1);
      unpack_w(rw[1][1], nhi[2], nhi[3], 1);
      SGLK_RD16(mlo[0], a_lo, 0);          SGLK_RD16(mhi[0], a_hi, 0);
      SGLK_RD4(raw[0], ts0, 0);
    } else {
    SGLK_RD16(nlo[0], b_lo, kNfImm[0]);  SGLK_RD16(nhi[0], b_hi, kNfImm[0]);
    SGLK_RD16(mlo[0], a_lo, 0);          SGLK_RD16(mhi[0], a_hi, 0);
    SGLK_RD4(raw[0], ts0, 0);
    SGLK_RD16(nlo[1], b_lo, kNfImm[1]);  SGLK_RD16(nhi[1], b_hi, kNfImm[1]);
    SGLK_RD16(nlo[2], b_lo, kNfImm[2]);  SGLK_RD16(nhi[2], b_hi, kNfImm[2]);
    SGLK_RD16(nlo[3], b_lo, kNfImm[3]);  SGLK_RD16(nhi[3], b_hi, kNfImm[3]);
    }
  }
  if (kDma) {
    dma_piece(cur_t, 1, 1, 0, 0);
    dma_piece(cur_t, 1, 1, 0, 1);
    dma_piece(cur_t, 1, 1, 0, 2);
  }

  for (; unit < n_units; ++unit) {
    const TileDesc nxt = describe(unit + 1);
    {
      const int kb = 0;
      SGLK_BLOCK(true, MS)
    }
    for (int kb = 1; kb < nkb; ++kb) SGLK_BLOCK(false, MS)
    prv = cur_t;
    cur_t = nxt;
  }
#undef SGLK_BLOCK
#undef SGLK_STEP
  // the reads and DMA issued by the last step have no consumer: drain them; promote the last m-step; store the last unit
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (kBW) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(cur[1][0]), "+v"(cur[1][1]), "+v"(cur[1][2]), "+v"(cur[1][3]));  // (MFMA -> VALU)
  } else {
    epi = load_epi(prv);
    // (the last MFMAs' results are read by compiler-scheduled VALU code: give the matrix pipe time to retire them)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) asm volatile("" : "+v"(acc[MS - 1][nf]));
  }
  SGLK_PROMOTE(MS - 1, 1, 0) SGLK_PROMOTE(MS - 1, 1, 1) SGLK_PROMOTE(MS - 1, 1, 2) SGLK_PROMOTE(MS - 1, 1, 3)
#pragma unroll
  for (int mf = 0; mf < MS; ++mf) store_rows(prv, acc[mf], mf, epi);
  if constexpr (PROBE == 5) {
    if (stamps != nullptr && gblk >= 60) stamps[((int64_t)blockIdx.x * 8 + wave) * 64 + lane] = stampv;  // (window complete)
  }
#undef SGLK_PROMOTE
#undef SGLK_MFMA
#undef SGLK_RD16
#undef SGLK_RD4
#undef SGLK_RDW
#undef SGLK_FRAG
}

template <typename OutT, int MODE, bool HW_SCALE, int PROBE, int MS>
__global__ __launch_bounds__(512) void gemm_8bit_persist_kernel(
    OutT* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ sa, const float* __restrict__ sb, const OutT* __restrict__ bias, int M, int N, int K,
    int64_t lda, int64_t ldb, int64_t ldc, int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, int tiles_m,
    int tiles_n, int all_halves, uint32_t* __restrict__ stamps, const void* __restrict__ x0,
    const void* __restrict__ x1) {
  __shared__ __attribute__((aligned(256))) char smem[kStages * kStageBytes];
  gemm_8bit_persist_phase<OutT, MODE, HW_SCALE, PROBE, MS>(smem, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, sa_sm, sa_sk,
                                                             sb_sk, sb_sn, tiles_m, tiles_n, all_halves, stamps, x0, x1);
}

// ONE launch for a problem whose last round of 256-row tiles is only partly filled (round 5; the fp8 block-scale kernel has
// run this way since round 4: one launch instead of a launch of whole tiles, a pipeline drain and a second launch of half
// tiles): a persistent workgroup runs its whole tiles (MS = 8), then - when its XCD's last round fits twice - one 128-row
// half tile of that round (MS = 4). Used by fp8_scaled_mm, int8_scaled_mm and the QServe W4A8 modes.
template <typename OutT, int MODE, bool HW_SCALE, int PROBE>
__global__ __launch_bounds__(512) void gemm_8bit_persist2_kernel(
    OutT* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ sa, const float* __restrict__ sb, const OutT* __restrict__ bias, int M, int N, int K,
    int64_t lda, int64_t ldb, int64_t ldc, int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, int tiles_m,
    int tiles_n, uint32_t* __restrict__ stamps, const void* __restrict__ x0, const void* __restrict__ x1) {
  __shared__ __attribute__((aligned(256))) char smem[kStages * kStageBytes];
  gemm_8bit_persist_phase<OutT, MODE, HW_SCALE, PROBE, 8>(smem, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, sa_sm, sa_sk,
                                                            sb_sk, sb_sn, tiles_m, tiles_n, 0, stamps, x0, x1);
  // (the phase has drained its own LDS reads and DMA; no wave may start the next prologue's DMA while another still reads
  //  the stages)
  asm volatile("s_barrier" ::: "memory");
  gemm_8bit_persist_phase<OutT, MODE, HW_SCALE, PROBE, 4>(smem, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, sa_sm, sa_sk,
                                                            sb_sk, sb_sn, tiles_m, tiles_n, 0, stamps, x0, x1);
}

// ---------------------------------------------------------------------------------------------------------
// fp8_blockwise_scaled_mm on the 32 x 32 x 64 form of the MX MFMA: the same persistent pipeline (tile walk, LDS image of
// a, LDS-DMA through buffer resources, one barrier per K block in front of its last m-step, the finished tile stored in
// the first K block of the next one), other arithmetic schedule. Measured on the register-only streams of
// tools/kbench: a v_mfma_scale_f32_16x16x128_f8f6f4 keeps the SIMD's vector issue for nearly all of its
// 32 cycles, so its four promotion FMAs, the LDS reads and the DMA pieces ADD to the matrix time; the 64-cycle 32x32x64
// form holds it for ~11, and the 16 FMAs of a 32 x 32 partial fit into the gap behind it.
//  * wave tile 64 (n) x 32 MS (m) as 2 n-fragments x MS m-fragments of 32 rows; lane (i = lane % 32, h = lane / 32) supplies
//    row i, 16-byte chunks 4 h + {0, 1} (first MFMA of a K block) and 4 h + {2, 3} (second) - the same for both operands,
//    so the MFMAs pair equal k. Both LDS tiles use a's key (chunk c of row r at c ^ ((r >> 1) & 7)): conflict-free for
//    ds_read_b128 over 32 rows of one chunk.
//  * an m-step is A1 B1 A2 B2 (A, B = the two n-fragments; 1, 2 = the two K halves; A2 accumulates onto A1 in cur0, B2 onto
//    B1 in cur1). cur0 is folded into the accumulators behind B2 (its A2 was issued a whole MFMA earlier), cur1 behind the
//    next step's A1: no FMA sits behind the MFMA it depends on, and every gap carries at most 16 FMAs.
//  * n-fragment row i of the MFMA is LDS row 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3) of its 32: a lane then owns 16
//    consecutive output columns of one row (two 16-byte stores per 32 x 32 tile).
typedef float v16f __attribute__((ext_vector_type(16)));

// One PHASE of the kernel below: the units of one kind (MS = 4: whole 256-row tiles, MS = 2: 128-row half tiles) of this
// workgroup, prologue to final stores. smem = the workgroup's kStages * kStageBytes of LDS.
template <typename OutT, int MS, int PROBE>  // MS m-steps (32 rows) per K block: 4 = 256-row tiles, 2 = 128-row half tiles
__device__ __forceinline__ void gemm_fp8bw_x32_phase(
    char* smem, OutT* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ sa, const float* __restrict__ sb, int M, int N, int K, int64_t lda, int64_t ldb,
    int64_t ldc, int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, int tiles_m, int tiles_n, int all_halves,
    uint32_t* __restrict__ stamps, int ksplit, int64_t slab) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // K split (OutT = float, few rows x deep K: see launch()): unit = tile x slice; slice s multiplies K blocks [s nkb, (s + 1) nkb)
  // and stores its fp32 partial tile into slab s of `out` (slab elements apart); a second kernel adds the slabs up.
  const int nkb = K / BK / ksplit;  // >= 3
  constexpr bool kDma = PROBE == 0 || PROBE >= 3, kStore = PROBE == 0 || PROBE == 1 || (PROBE >= 12 && PROBE <= 19);
  // LDS stage of this phase: a rows [64 MS][128 B], b^T rows [256][128 B], 256 row scales (+ 1 KiB spare). Whole tiles: two
  // stages of 66 KiB. Half tiles: THREE stages of 50 KiB - with 1024 cycles of MFMAs per K block and the block's data
  // requested one block ahead, the half-tile loop ran at the latency of its LDS-DMA (2450 shader cycles per block,
  // in-kernel stamps); here block g + 2's b pieces go out in block g and block g + 3's a pieces behind its barrier.
  constexpr int NST = MS == 2 ? 3 : 2;
  constexpr int kOffB = MS * 64 * BK, kOffS = kOffB + kTileBytes, kStg = kOffS + 2048;
  static_assert(NST * kStg <= 160 * 1024 - 8192, "LDS");
#ifdef SGLK_PROBES
  // (diagnostic build, stamps != nullptr) shader cycles and 100 MHz ticks of this phase -> the clock it ran at
  const uint64_t st_c0 = stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
  // (diagnostic build) 5: every DMA piece out of range (issue + LDS write of zeros, no fetch); 6: a / b pieces as plain
  // loads into registers (fetch, no LDS write); results are garbage, no stores

  // ---- this workgroup's units: the tile walk of gemm_8bit_persist_kernel
  constexpr bool kSlabs = sizeof(OutT) == 4;
  const int nt = tiles_m * tiles_n * ksplit;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int q8 = nt >> 3, rem = nt & 7;
  const int run_first = xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8;
  const int run_len = q8 + (xcd < rem ? 1 : 0);
  const int nblk_max = (N + 127) / 128 - 1;
  const int rounds = run_len / slots, left = run_len - rounds * slots;
  const bool split = left > 0 && 2 * left <= slots;
  constexpr bool kHalf = MS == 2;
  const int n_units = (kHalf && all_halves) ? (2 * run_len > slot ? (2 * run_len - slot + slots - 1) / slots : 0)
                      : kHalf             ? ((split && slot < 2 * left) ? 1 : 0)
                                          : (split ? rounds : rounds + (slot < left ? 1 : 0));
  if (n_units == 0) return;

  auto describe = [&](int unit) -> TileDesc {  // unit >= n_units: the null tile
    TileDesc d;
    bool live = unit < n_units;
    const int ht = slot + unit * slots;
    const int local = !live ? 0 : (kHalf && all_halves) ? (ht >> 1) : kHalf ? rounds * slots + (slot >> 1) : slot + unit * slots;
    const int lower = all_halves ? (ht & 1) : (slot & 1);
    const int tile_s = run_first + local;
    const int tile = kSlabs ? tile_s / ksplit : tile_s;
    const int ks = kSlabs ? tile_s - tile * ksplit : 0;
    constexpr int GM = 4;
    const int group = tile / (GM * tiles_n);
    const int first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_group = tile - group * GM * tiles_n;
    const int tm = __builtin_amdgcn_readfirstlane(first_m + in_group % gsz);
    const int tn = __builtin_amdgcn_readfirstlane(in_group / gsz);
    const int trows = kHalf ? BM / 2 : BM;
    const int m0 = tm * BM + ((kHalf && lower) ? BM / 2 : 0), n0 = tn * BN;
    const int rows_a = (M - m0) < trows ? (M - m0) : trows, rows_b = (N - n0) < BN ? (N - n0) : BN;
    live = live && rows_a > 0;
    d.ncols = rows_b;
    d.wrows = MS * 32;
    d.m0 = m0;
    d.n0 = n0;
    const int kb0 = ks * nkb;  // the slice's first K block
    d.pa = a + (int64_t)m0 * lda + kb0 * BK;
    d.pb = b + (int64_t)n0 * ldb + kb0 * BK;
    d.ps = sa + (int64_t)m0 * sa_sm + (int64_t)kb0 * sa_sk;
    d.po = (void*)(out + (kSlabs ? (int64_t)ks * slab : 0) + (int64_t)m0 * ldc + n0);
    d.nrec_a = (live && PROBE != 5) ? (uint32_t)((int64_t)(rows_a - 1) * lda + nkb * BK) : 0u;
    d.nrec_b = (live && PROBE != 5) ? (uint32_t)((int64_t)(rows_b - 1) * ldb + nkb * BK) : 0u;
    d.nrec_s = (live && wave < 4) ? (uint32_t)(((int64_t)(rows_a - 1) * sa_sm + (int64_t)(nkb - 1) * sa_sk + 1) * 4) : 0u;
    d.nrec_o = (live && kStore) ? (uint32_t)(((int64_t)(rows_a - 1) * ldc + rows_b) * (int64_t)sizeof(OutT)) : 0u;
    int nblk = (n0 + wn * 64) >> 7;
    nblk = nblk < nblk_max ? nblk : nblk_max;
    d.sbw = sb + (int64_t)nblk * sb_sn + (int64_t)kb0 * sb_sk;
    // (probe 16: consecutive units of a workgroup share their a panel - the tile index advances by the slot count, a multiple of
    //  the group of four m-tiles; odd units walk K backwards, so the panel's last blocks are re-read while the XCD's L2 holds them)
    d.rev = (PROBE == 16) ? (unit & 1) : 0;
    return d;
  };
  auto pick = [](bool c, const TileDesc& x, const TileDesc& y) -> TileDesc {  // scalar selects
    TileDesc d;
    d.pa = c ? x.pa : y.pa;  d.pb = c ? x.pb : y.pb;  d.ps = c ? x.ps : y.ps;  d.sbw = c ? x.sbw : y.sbw;
    d.po = c ? x.po : y.po;
    d.nrec_a = c ? x.nrec_a : y.nrec_a;  d.nrec_b = c ? x.nrec_b : y.nrec_b;  d.nrec_s = c ? x.nrec_s : y.nrec_s;
    d.nrec_o = c ? x.nrec_o : y.nrec_o;  d.ncols = c ? x.ncols : y.ncols;  d.wrows = c ? x.wrows : y.wrows;
    d.m0 = c ? x.m0 : y.m0;  d.n0 = c ? x.n0 : y.n0;
    d.rev = c ? x.rev : y.rev;
    return d;
  };

  const uint32_t lds_base = (uint32_t)(uintptr_t)SGLK_LDS(smem);
  // DMA piece p of a tile = rows 8p..8p+7; lane -> row 8p + lane/8, chunk (lane%8) ^ key(row), key = (row>>1)&7 =
  // (4 (p&1) + lane/16) & 7 for both tiles
  uint32_t voff_a[2], voff_b[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const uint32_t ch = (uint32_t)(((lane & 7) ^ ((par * 4 + (lane >> 4)) & 7)) << 4);
    voff_a[par] = (uint32_t)(lane >> 3) * (uint32_t)lda + ch;
    voff_b[par] = (uint32_t)(lane >> 3) * (uint32_t)ldb + ch;
  }
  const uint32_t voff_s = (uint32_t)tid * (uint32_t)sa_sm * 4u;
  // one 1-KiB piece per call: part 0, 1 = rows of a, part 2, 3 = rows of b^T (sub 0, 1 each); (part 0, sub 2) = row scales
  auto dma_piece = [&](const TileDesc& d, int kb_, int s, int part, int sub) {
    const int kb = PROBE == 4 ? 0 : (PROBE == 16 && d.rev) ? nkb - 1 - kb_ : kb_;
    char* base = smem + s * kStg;
    if (sub == 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(d.ps, d.nrec_s), SGLK_LDS(base + kOffS + wave * 256), 4,
                                               voff_s, kb * (int)sa_sk * 4, 0, 0);
      return;
    }
    // (half tiles: a has 128 rows = 16 pieces, two per wave - part 0 only)
    const int ii = (MS == 2 && part < 2) ? sub : (part & 1) * 2 + sub, piece = (MS == 2 && part < 2) ? wave * 2 + sub : wave * 4 + ii;
    if constexpr (PROBE == 6) {
      const v4i t = __builtin_amdgcn_raw_buffer_load_b128(part < 2 ? make_rsrc(d.pa, d.nrec_a) : make_rsrc(d.pb, d.nrec_b),
                                                          part < 2 ? voff_a[ii & 1] : voff_b[ii & 1],
                                                          kb * BK + piece * 8 * (int)(part < 2 ? lda : ldb), 0);
      asm volatile("" ::"v"(t));
      return;
    }
    if (part < 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(d.pa, d.nrec_a), SGLK_LDS(base + piece * 1024), 16,
                                               voff_a[ii & 1], kb * BK + piece * 8 * (int)lda, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(d.pb, d.nrec_b), SGLK_LDS(base + kOffB + piece * 1024), 16,
                                               voff_b[ii & 1], kb * BK + piece * 8 * (int)ldb, 0, 0);
    }
  };

  // fragment addressing (byte offsets inside a stage)
  const int li = lane & 31, lh = lane >> 5;
  // LDS row of MFMA row li inside a 32-row n-fragment: register v of lane (i, h) is column 16 (v / 8) + 8 h + v % 8, so that
  // one store instruction writes 32 contiguous bytes per row (the two h lanes side by side)
  const int brow = (li >> 4) * 16 + ((li >> 2) & 1) * 8 + ((li >> 3) & 1) * 4 + (li & 3);
  const int frag_off_a = li * 128 + (((4 * lh) ^ ((li >> 1) & 7)) << 4);        // chunk 4h; chunks 4h + q at ^ (q << 4)
  const int frag_off_b = brow * 128 + (((4 * lh) ^ ((brow >> 1) & 7)) << 4);

  // stores: lane (i, h) owns row wm * 32 MS + 32 mf + i, columns wn * 64 + 32 nf + 16 h .. + 15
  const uint32_t orow_off = (uint32_t)(((int64_t)(wm * (MS * 32) + li) * ldc + wn * 64 + lh * 8) * (int64_t)sizeof(OutT));
  // (Round 5, measured and left as probe 15 / kbench variant 37: the rows of a whole tile staged through LDS - 16 rows of the
  // wave's 64 columns in the 18 KiB the two whole-tile stages leave free, row stride 144 B, read back a row per eight lanes - so
  // that a store instruction writes eight whole 128-byte lines instead of a quarter of 32. The MLA epilogue of this round gained
  // 2.5x from that change; here it is a wash - 0.2395 against 0.2396 ms at (4096, 14336, 4096), 0.5026 / 0.5046 at 8192^3,
  // interleaved on one box - the stores ride under the next tile's MFMAs either way, and the staged form costs the prologues
  // three spilled registers. The release kernel keeps the direct stores.)
  constexpr bool kStaged = MS == 4 && PROBE == 15 && sizeof(OutT) == 2;
  auto store_frag = [&](const TileDesc& d, const float (&accm)[2][16], int mf) {
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(d.po, d.nrec_o);
    if constexpr (kStaged) {
      static_assert(2 * kStg + 8 * 2304 <= 3 * (kStageBytes - kTileBytes / 2), "the staging rows fit behind the two whole-tile stages");
      // (addresses from a laundered lane id, formed here: the K loop has no register to carry them - 256 per wave, all taken)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int li2 = ln & 31, lh2 = ln >> 5;
      const uint32_t stg_w = lds_base + (uint32_t)(2 * kStg + wave * 2304 + (li2 & 15) * 144 + lh2 * 16);
      const uint32_t stg_r = lds_base + (uint32_t)(2 * kStg + wave * 2304 + (ln >> 3) * 144 + (ln & 7) * 16);
      const uint32_t vo2 = (wn * 64 + (ln & 7) * 8 < d.ncols)
                               ? (uint32_t)(((int64_t)(wm * (MS * 32) + (ln >> 3)) * ldc + wn * 64 + (ln & 7) * 8) * (int64_t)sizeof(OutT))
                               : 0x80000000u;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
          for (int hv = 0; hv < 2; ++hv) {
            Vec<OutT, 8> v;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (OutT)accm[nf][hv * 8 + c];
            const v4i dq = __builtin_bit_cast(v4i, v);
            if ((li2 >> 4) == half)  // (this half's sixteen rows: 32 lanes write their 16-byte piece)
              asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(stg_w), "v"(dq), "n"(nf * 64 + hv * 32) : "memory");
          }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          v4i rr;
          if (r2 == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rr) : "v"(stg_r) : "memory");
          else asm volatile("ds_read_b128 %0, %1 offset:1152\n\ts_waitcnt lgkmcnt(0)" : "=v"(rr) : "v"(stg_r) : "memory");
          const int soff2 = __builtin_amdgcn_readfirstlane((mf * 32 + half * 16 + r2 * 8) * (int)ldc * (int)sizeof(OutT));
          __builtin_amdgcn_raw_buffer_store_b128(rr, ro, (int)vo2, soff2, 0);
          asm volatile("s_nop 4" ::"v"(rr));  // (store data is read for a few cycles after issue)
        }
      }
      return;
    }
    const int soff = __builtin_amdgcn_readfirstlane(mf * 32 * (int)ldc * (int)sizeof(OutT));
    if constexpr (kSlabs) {  // fp32 partial tile of a K slice: the lane's 8 columns are two 16-byte stores
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
          const uint32_t vo = (wn * 64 + nf * 32 + hv * 16 + lh * 8 < d.ncols)
                                  ? orow_off + (uint32_t)((nf * 32 + hv * 16) * 4) : 0x80000000u;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float f0 = accm[nf][hv * 8 + q * 4 + 0], f1 = accm[nf][hv * 8 + q * 4 + 1], f2 = accm[nf][hv * 8 + q * 4 + 2],
                        f3 = accm[nf][hv * 8 + q * 4 + 3];
            const v4i data = {(int)__float_as_uint(f0), (int)__float_as_uint(f1), (int)__float_as_uint(f2), (int)__float_as_uint(f3)};
            __builtin_amdgcn_raw_buffer_store_b128(data, ro, (int)vo + q * 16, soff, 0);
            asm volatile("s_nop 4" ::"v"(data));
          }
        }
      return;
    } else {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int hv = 0; hv < 2; ++hv) {
        Vec<OutT, 8> v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (OutT)accm[nf][hv * 8 + c];
        const uint32_t vo = (wn * 64 + nf * 32 + hv * 16 + lh * 8 < d.ncols)
                                ? orow_off + (uint32_t)((nf * 32 + hv * 16) * (int)sizeof(OutT)) : 0x80000000u;
        const v4i data = __builtin_bit_cast(v4i, v);
        __builtin_amdgcn_raw_buffer_store_b128(data, ro, (int)vo, soff, PROBE == 12 ? 2 : PROBE == 13 ? 3 : PROBE == 14 ? 17 : 0);  /* probes 12..14: nt / nt + sc0 / sc0 sc1 stores - 0.306 / 0.307 / 0.244 ms against 0.233 with the default policy (write-back through L2) */
        asm volatile("s_nop 4" ::"v"(data));  // (store data is read for a few cycles after issue: see the kernel above)
      }
    }
  };

  float acc[MS][2][16];
#pragma unroll
  for (int mf = 0; mf < MS; ++mf)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

// (diagnostic build, garbage results: probe 17 = no promotion FMAs, 18 = no LDS fragment reads, 19 = neither - what each costs
//  under the power cap: time x in-kernel clock, kbench gemm with GEMM_CLOCK=1)
#define X32_RD16(dst, addr, imm)                                                                   \
  if constexpr (PROBE != 18 && PROBE != 19) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm)); \
  else asm volatile("" : "=v"(dst))
#define X32_RD4(dst, addr, imm) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define X32_FRAG(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)
// first / second MFMA of a 32 x 32 partial (K halves s = 0, 1)
#define X32_MFMA1(cur, nf)                                                                                     \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"                      \
               : "=&v"(cur)                                                                                    \
               : "v"(X32_FRAG(nq[nf][0][0], nq[nf][0][1])), "v"(X32_FRAG(mq[0][0], mq[0][1])), "v"(one_e8m0));
#define X32_MFMA2(cur, nf)                                                                                     \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"                     \
               : "+v"(cur)                                                                                     \
               : "v"(X32_FRAG(nq[nf][1][0], nq[nf][1][1])), "v"(X32_FRAG(mq[1][0], mq[1][1])), "v"(one_e8m0));
// acc[row][nf][:] += cur[:] * sc
#define X32_PROMOTE(row, nf, cur)                                                                              \
  if constexpr (PROBE != 17 && PROBE != 19)                                                                    \
  _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_)                                                            \
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[row][nf][r_]) : "v"(cur[r_]), "v"(sc));

  v4i nq[2][2][2];  // [n-fragment][K half][chunk]
  v4i mq[2][2];     // [K half][chunk] of the running m-fragment (each half is re-read right behind its last MFMA)
  float raw, sbv;
  v16f cur0, cur1;
  float sc = 0.f;   // row scale x column-block scale of the partial that is folded next
#pragma unroll
  for (int r = 0; r < 16; ++r) cur1[r] = 0.f;  // (the first step folds nothing)
  int one_e8m0 = 127;
  asm volatile("" : "+v"(one_e8m0), "+v"(sc), "+v"(cur1));
  int gblk = 0, stg = 0;  // K blocks done; (three stages) the stage of the running block

  // m-step mf of a K block. LAST: behind the block's barrier; its gaps carry the reads of the next block's fragments.
  // LDS reads in issue order - a step: m half 0 (2), m half 1 (2), row scale; the last step: n0 half 0 (2), n1 half 0 (2),
  // m half 0 (2), n0 half 1 (2), n1 half 1 (2), m half 1 (2), row scale.
#define X32_STEP(mf, STORE, LAST)                                                                              \
  {                                                                                                            \
    constexpr int prow_ = ((mf) + MS - 1) % MS;                                                                \
    if ((mf) == 0) {                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(nq[0][0][0]), "+v"(nq[0][0][1]), "+v"(nq[1][0][0]), "+v"(nq[1][0][1]), \
                                            "+v"(mq[0][0]), "+v"(mq[0][1]));                                   \
    } else if (!(LAST)) {                                                                                      \
      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(mq[0][0]), "+v"(mq[0][1]));                                   \
    }                                                                                                          \
    X32_MFMA1(cur0, 0)                                                                                         \
    if (LAST) { X32_RD16(nq[0][0][0], nb0, 0); X32_RD16(nq[0][0][1], nb1, 0); }                                \
    asm volatile("s_nop 2" : "+v"(cur1)); /* (B2 of the step before is >= 18 passes old when the first FMA issues) */ \
    X32_PROMOTE(prow_, 1, cur1)                                                                                \
    X32_MFMA1(cur1, 1)                                                                                         \
    if (LAST) {                                                                                                \
      X32_RD16(nq[1][0][0], nb0, 4096);  X32_RD16(nq[1][0][1], nb1, 4096);                                     \
      X32_RD16(mq[0][0], na0, 0);        X32_RD16(mq[0][1], na1, 0);                                           \
      if (kDma) dma_piece(dL, kbL, sL, 0, 0);                                                                  \
    } else {                                                                                                   \
      X32_RD16(mq[0][0], a0, ((mf) + 1) * 4096);  X32_RD16(mq[0][1], a1, ((mf) + 1) * 4096);                   \
      if (kDma && MS == 4 && (mf) == 0) dma_piece(dE, kbE, sE, 1, 0);                                          \
      if (kDma && MS == 4 && (mf) == 1) dma_piece(dE, kbE, sE, 2, 1);                                          \
      if (kDma && MS == 2) dma_piece(dE, kbE, sE, 2, 0);                                                       \
    }                                                                                                          \
    if constexpr (STORE) {                                                                                     \
      store_frag(prv, acc[mf], (mf));                                                                          \
      _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f; \
      asm volatile("" : "+v"(acc[mf][0][0]), "+v"(acc[mf][1][0]));                                             \
    }                                                                                                          \
    if (!(LAST)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(mq[1][0]), "+v"(mq[1][1]), "+v"(raw));             \
    X32_MFMA2(cur0, 0)                                                                                         \
    const float scn_ = raw * sbv;                                                                              \
    if (LAST) {                                                                                                \
      X32_RD16(nq[0][1][0], nb2, 0);  X32_RD16(nq[0][1][1], nb3, 0);                                           \
      if (kDma) dma_piece(dL, kbL, sL, 0, 1);                                                                  \
    } else {                                                                                                   \
      /* (round 4: with no two pieces back to back - the second of each pair moved behind the step's last MFMA - 0.2289 against */ \
      /* 0.2268 ms, no gain: between 64-cycle MFMAs the pair's issue is covered)                                                   */ \
      /* (whole tiles: the block's last pieces go out two m-steps before its barrier: issued in the step in front of it */ \
      /* they were still in flight at the vmcnt(0) there: +54 us at the headline shape)                                  */ \
      if (kDma && MS == 4 && (mf) == 0) { dma_piece(dE, kbE, sE, 1, 1); dma_piece(dE, kbE, sE, 2, 0); }        \
      if (kDma && MS == 4 && (mf) == 1) { dma_piece(dE, kbE, sE, 3, 0); dma_piece(dE, kbE, sE, 3, 1); }        \
      if (kDma && MS == 2) { dma_piece(dE, kbE, sE, 2, 1); dma_piece(dE, kbE, sE, 3, 0); }                     \
    }                                                                                                          \
    X32_MFMA2(cur1, 1)                                                                                         \
    if (LAST) {                                                                                                \
      X32_RD16(nq[1][1][0], nb2, 4096);  X32_RD16(nq[1][1][1], nb3, 4096);                                     \
      X32_RD16(mq[1][0], na2, 0);        X32_RD16(mq[1][1], na3, 0);                                           \
      X32_RD4(raw, nts, 0);                                                                                    \
      if (kDma) dma_piece(dL, kbL, sL, 0, 2);                                                                  \
    } else {                                                                                                   \
      X32_RD16(mq[1][0], a2, ((mf) + 1) * 4096);  X32_RD16(mq[1][1], a3, ((mf) + 1) * 4096);                   \
      X32_RD4(raw, ts_addr, ((mf) + 1) * 128);                                                                 \
      if (kDma && MS == 2) dma_piece(dE, kbE, sE, 3, 1);                                                       \
    }                                                                                                          \
    sc = scn_;                                                                                                 \
    asm volatile("" : "+v"(sc), "+v"(cur0));                                                                   \
    X32_PROMOTE((mf), 0, cur0)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
// WHERE: 0 = any block (the tiles of blocks + 1 / + 2 / + 3 picked by scalar selects: the first block of a tile), 1 = those
// blocks inside the running tile (the steady loop: no selects between its MFMAs - with them the compiler's placement of ~40
// scalar instructions differed between two instantiations of this very kernel by 5.6 % of the GEMM's time), 2 = the tile's last
// but one block, 3 = its last block (whole tiles), 4 = 1 right behind the store block (half tiles: its wait counts the stores).
// dE / kbE / sE: tile, K block and stage of the pieces that go out in front of the barrier (block + 1, with three stages
// block + 2); dL / kbL / sL: of those behind it (block + 2 into the stage just read; with three stages block + 3).
#define X32_BLOCK(STORE, WHERE)                                                                                \
  {                                                                                                            \
    const int s = NST == 2 ? (gblk & 1) : stg;                                                                 \
    const int s1 = NST == 2 ? (s ^ 1) : (s == 2 ? 0 : s + 1), s2 = NST == 2 ? s : (s1 == 2 ? 0 : s1 + 1);      \
    const uint32_t sbase = lds_base + (uint32_t)(s * kStg);                                                    \
    const uint32_t nbase = lds_base + (uint32_t)(s1 * kStg);                                                   \
    constexpr bool steady_ = (WHERE) == 1 || (WHERE) == 4;                                                     \
    const bool in1 = (WHERE) == 0 ? kb + 1 < nkb : (WHERE) != 3, in2 = (WHERE) == 0 ? kb + 2 < nkb : steady_,  \
               in3 = (WHERE) == 0 ? kb + 3 < nkb : steady_;                                                    \
    const TileDesc d1 = (WHERE) == 0 ? pick(in1, cur_t, nxt) : (WHERE) == 3 ? nxt : cur_t;                     \
    const TileDesc d2 = (WHERE) == 0 ? pick(in2, cur_t, nxt) : steady_ ? cur_t : nxt;                          \
    const TileDesc d3 = NST == 2 ? d2 : (WHERE) == 0 ? pick(in3, cur_t, nxt) : steady_ ? cur_t : nxt;          \
    const int kb1 = in1 ? kb + 1 : 0, kb2 = in2 ? kb + 2 : kb + 2 - nkb, kb3 = in3 ? kb + 3 : kb + 3 - nkb;    \
    const TileDesc& dE = NST == 2 ? d1 : d2;                                                                   \
    const TileDesc& dL = NST == 2 ? d2 : d3;                                                                   \
    const int kbE = NST == 2 ? kb1 : kb2, kbL = NST == 2 ? kb2 : kb3, sE = NST == 2 ? s1 : s2, sL = s;         \
    uint32_t a0, a1, a2, a3, ts_addr, nb0 = 0, nb1 = 0, nb2 = 0, nb3 = 0, na0 = 0, na1 = 0, na2 = 0, na3 = 0, nts = 0; \
    {                                                                                                          \
      int fo = frag_off_a;                                                                                     \
      asm volatile("" : "+v"(fo));                                                                             \
      a0 = sbase + (uint32_t)(wm * (MS * 32) * 128) + (uint32_t)fo;                                            \
      a1 = a0 ^ 16u;  a2 = a0 ^ 32u;  a3 = a0 ^ 48u;                                                           \
      ts_addr = sbase + (uint32_t)kOffS + (uint32_t)(wm * (MS * 32) * 4) + (uint32_t)((fo >> 7) << 2);         \
    }                                                                                                          \
    if constexpr (MS == 4) { X32_STEP(0, STORE, false) X32_STEP(1, STORE, false) X32_STEP(2, STORE, false) }   \
    else { X32_STEP(0, STORE, false) }                                                                         \
    float sbv_next = d1.sbw[(int64_t)((PROBE == 16 && d1.rev) ? nkb - 1 - kb1 : kb1) * sb_sk];                  \
    /* (round 3, probe 15 - counted vmcnt at the store block's barrier of whole tiles so that the stores get one more K block */ \
    /* before anything waits - changed nothing, 0.2376 against 0.2378 ms: the 33 MB burst takes four K blocks to drain)          */ \
    /* Half tiles, three stages: vmcnt retires in issue order; younger than block + 1's pieces are the 3 pieces issued behind    */ \
    /* the last barrier and the 4 issued in this block's first step - plus the 4 stores of a store block's step, here or in the  */ \
    /* block before.                                                                                                             */ \
    if constexpr (NST == 3 && ((STORE) || (WHERE) == 4)) {                                                     \
      asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)\n\ts_barrier"                                               \
                   : "+v"(mq[0][0]), "+v"(mq[0][1]), "+v"(mq[1][0]), "+v"(mq[1][1]), "+v"(raw)                  \
                   :                                                                                           \
                   : "memory");                                                                                \
    } else if constexpr (NST == 3) {                                                                           \
      asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier"                                                \
                   : "+v"(mq[0][0]), "+v"(mq[0][1]), "+v"(mq[1][0]), "+v"(mq[1][1]), "+v"(raw)                  \
                   :                                                                                           \
                   : "memory");                                                                                \
    } else {                                                                                                   \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"                                                \
                   : "+v"(mq[0][0]), "+v"(mq[0][1]), "+v"(mq[1][0]), "+v"(mq[1][1]), "+v"(raw)                  \
                   :                                                                                           \
                   : "memory");                                                                                \
    }                                                                                                          \
    asm volatile("" : "+v"(sbv_next));                                                                         \
    /* (timing probes 8 / 10, no stores: 64 .. 128 cycles of s_sleep for one half of the waves behind the barrier, to put the */ \
    /* two waves of a SIMD half an m-step apart - no effect: 0.2205 .. 0.2225 ms with or without, against 0.2394 with stores)  */ \
    if (PROBE == 10 && wave < 4) __builtin_amdgcn_s_sleep(1);                                                  \
    if (PROBE == 8 && wave >= 4) __builtin_amdgcn_s_sleep(2);                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    {                                                                                                          \
      int foa = frag_off_a, fob = frag_off_b;                                                                  \
      asm volatile("" : "+v"(foa), "+v"(fob));                                                                 \
      nb0 = nbase + (uint32_t)(kOffB + wn * 64 * 128) + (uint32_t)fob;                                         \
      nb1 = nb0 ^ 16u;  nb2 = nb0 ^ 32u;  nb3 = nb0 ^ 48u;                                                     \
      na0 = nbase + (uint32_t)(wm * d1.wrows * 128) + (uint32_t)foa;                                           \
      na1 = na0 ^ 16u;  na2 = na0 ^ 32u;  na3 = na0 ^ 48u;                                                     \
      nts = nbase + (uint32_t)kOffS + (uint32_t)(wm * d1.wrows * 4) + (uint32_t)((foa >> 7) << 2);             \
      asm volatile("" : "+v"(nb0), "+v"(nb1), "+v"(nb2), "+v"(nb3), "+v"(na0), "+v"(na1), "+v"(na2), "+v"(na3), "+v"(nts)); \
    }                                                                                                          \
    X32_STEP(MS - 1, STORE, true)                                                                              \
    sbv = sbv_next;                                                                                            \
    ++gblk;                                                                                                    \
    stg = s1;                                                                                                  \
  }

  int unit = 0;
  TileDesc cur_t = describe(0);
  TileDesc prv = describe(n_units);  // the null tile: nothing to store yet
  // ---- prologue: block 0 of the first unit lands, its fragments are read (in the last step's order), part 0 of block 1 goes out
#pragma unroll
  for (int part = 0; part < 4; ++part) {
    if (MS == 2 && part == 1) continue;
    dma_piece(cur_t, 0, 0, part, 0);
    dma_piece(cur_t, 0, 0, part, 1);
  }
  dma_piece(cur_t, 0, 0, 0, 2);
  sbv = cur_t.sbw[(PROBE == 16 && cur_t.rev) ? (int64_t)(nkb - 1) * sb_sk : 0];
  if constexpr (NST == 3) {  // all of block 1 and the a pieces + row scales of block 2 go out before anything waits
    if (kDma) {
      dma_piece(cur_t, 1, 1, 0, 0);  dma_piece(cur_t, 1, 1, 0, 1);
      dma_piece(cur_t, 1, 1, 2, 0);  dma_piece(cur_t, 1, 1, 2, 1);
      dma_piece(cur_t, 1, 1, 3, 0);  dma_piece(cur_t, 1, 1, 3, 1);
      dma_piece(cur_t, 1, 1, 0, 2);
      dma_piece(cur_t, 2, 2, 0, 0);  dma_piece(cur_t, 2, 2, 0, 1);
      dma_piece(cur_t, 2, 2, 0, 2);
      asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  asm volatile("" : "+v"(sbv));
  {
    const uint32_t b0 = lds_base + (uint32_t)(kOffB + wn * 64 * 128) + (uint32_t)frag_off_b;
    const uint32_t p0 = lds_base + (uint32_t)(wm * cur_t.wrows * 128) + (uint32_t)frag_off_a;
    const uint32_t ts0 = lds_base + (uint32_t)kOffS + (uint32_t)(wm * cur_t.wrows * 4) + (uint32_t)(li << 2);
    const uint32_t b1 = b0 ^ 16u, b2 = b0 ^ 32u, b3 = b0 ^ 48u, p1 = p0 ^ 16u, p2 = p0 ^ 32u, p3 = p0 ^ 48u;
    X32_RD16(nq[0][0][0], b0, 0);     X32_RD16(nq[0][0][1], b1, 0);
    X32_RD16(nq[1][0][0], b0, 4096);  X32_RD16(nq[1][0][1], b1, 4096);
    X32_RD16(mq[0][0], p0, 0);        X32_RD16(mq[0][1], p1, 0);
    X32_RD16(nq[0][1][0], b2, 0);     X32_RD16(nq[0][1][1], b3, 0);
    X32_RD16(nq[1][1][0], b2, 4096);  X32_RD16(nq[1][1][1], b3, 4096);
    X32_RD16(mq[1][0], p2, 0);        X32_RD16(mq[1][1], p3, 0);
    X32_RD4(raw, ts0, 0);
  }
  if (kDma && NST == 2) {
    dma_piece(cur_t, 1, 1, 0, 0);
    dma_piece(cur_t, 1, 1, 0, 1);
    dma_piece(cur_t, 1, 1, 0, 2);
  }

  for (; unit < n_units; ++unit) {
    const TileDesc nxt = describe(unit + 1);
    {
      const int kb = 0;
      X32_BLOCK(true, 0)
    }
    if constexpr (PROBE == 11) {  // (the round's first form: every block picks its tiles)
      for (int kb = 1; kb < nkb; ++kb) X32_BLOCK(false, 0)
    } else if constexpr (NST == 3) {  // (three stages: the tile's last three blocks pick the tiles of the pieces they issue)
      int kb = 1;
      if (kb < nkb - 3) {
        X32_BLOCK(false, 4)
        ++kb;
      }
      for (; kb < nkb - 3; ++kb) X32_BLOCK(false, 1)
      for (; kb < nkb; ++kb) X32_BLOCK(false, 0)
    } else {
      int kb = 1;
      for (; kb < nkb - 2; ++kb) X32_BLOCK(false, 1)
      if (kb == nkb - 2) {
        X32_BLOCK(false, 2)
        ++kb;
      }
      if (kb == nkb - 1) X32_BLOCK(false, 3)
    }
    prv = cur_t;
    cur_t = nxt;
  }
#undef X32_BLOCK
#undef X32_STEP
  // drain the reads and DMA of the last step; fold the last partial; store the last unit
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(cur1));  // (MFMA -> VALU)
  X32_PROMOTE(MS - 1, 1, cur1)
#pragma unroll
  for (int mf = 0; mf < MS; ++mf) store_frag(prv, acc[mf], mf);
#ifdef SGLK_PROBES
  if (stamps != nullptr && tid == 0) {
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    stamps[(blockIdx.x + (MS == 2 ? 256 : 0)) * 4 + 0] = (uint32_t)(c1 - st_c0);
    stamps[(blockIdx.x + (MS == 2 ? 256 : 0)) * 4 + 1] = (uint32_t)(r1 - st_r0);
    stamps[(blockIdx.x + (MS == 2 ? 256 : 0)) * 4 + 2] = (uint32_t)(n_units * nkb);
    stamps[(blockIdx.x + (MS == 2 ? 256 : 0)) * 4 + 3] = MS;
  }
#endif
#undef X32_PROMOTE
#undef X32_MFMA1
#undef X32_MFMA2
#undef X32_RD16
#undef X32_RD4
#undef X32_FRAG
}

// The kernel: ONE launch per GEMM. A persistent workgroup (one per CU) runs two phases - its whole 256-row tiles
// (MS = 4) and, when its XCD's last round is only partly filled, one 128-row half tile of that round (MS = 2) - so that
// 896 tiles on 256 CUs (3.5 rounds) cost 3.5 rounds of every CU instead of a launch of three rounds, a pipeline drain,
// and a second launch for the half round (round 3: 182 + 47 us). `half_first` picks the workgroups that run their half tile
// BEFORE their whole tiles: their tile boundaries then lie half a tile away from the others', and the 128 KiB a
// workgroup writes at each boundary no longer leave all 256 CUs within the same K block (the store burst of DESIGN 4.10).
// all_halves: every tile as two half tiles (few rows: twice the workgroups), the MS = 2 phase only.
// stagger: 0 = nobody, 1 = the upper half of an XCD's slots, 2 = every other pair of slots runs its half tile first,
// 3 = every workgroup of the odd XCDs (the workgroups that share panels in an L2 stay in step), 4 = everybody.
// Measured at (4096, 14336, 4096) in interleaved rounds on one device (kbench gemmab): 0.2200 / 0.2225 / 0.2250 ms for
// 0 / 1 / 2 - what the halved burst gives, the XCD's L2 takes back (16 whole + 16 half tiles in flight share 4 + 4 + 2
// panels in two K positions instead of 32 tiles sharing 4 + 8 in one): the release library uses 0.
template <typename OutT, int PROBE>
__global__ __launch_bounds__(512) void gemm_fp8bw_x32_kernel(
    OutT* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ sa, const float* __restrict__ sb, int M, int N, int K, int64_t lda, int64_t ldb,
    int64_t ldc, int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, int tiles_m, int tiles_n, int all_halves,
    int stagger, uint32_t* __restrict__ stamps, int ksplit, int64_t slab) {
  __shared__ __attribute__((aligned(256))) char smem[3 * (kStageBytes - kTileBytes / 2)];  // half tiles: 3 x 50 KiB; whole tiles 2 x 66 KiB
  const int slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const bool half_first = all_halves || (stagger == 1 ? 2 * slot >= slots : stagger == 2 ? ((slot >> 1) & 1) != 0
                                         : stagger == 3 ? (blockIdx.x & 1) != 0 : stagger == 4);
#pragma nounroll
  for (int ph = 0; ph < (all_halves ? 1 : 2); ++ph) {
    if ((ph == 0) == half_first) {
      gemm_fp8bw_x32_phase<OutT, 2, PROBE>(smem, out, a, b, sa, sb, M, N, K, lda, ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn,
                                          tiles_m, tiles_n, all_halves, stamps, ksplit, slab);
    } else {
      gemm_fp8bw_x32_phase<OutT, 4, PROBE>(smem, out, a, b, sa, sb, M, N, K, lda, ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn,
                                          tiles_m, tiles_n, 0, stamps, ksplit, slab);
    }
    // (the phase has drained its own LDS reads and DMA; no wave may start the next prologue's DMA while another still
    // reads the stages)
    asm volatile("s_barrier" ::: "memory");
  }
}

// ---------------------------------------------------------------------------------------------------------
// Kernel for few rows (decode: M <= 128; above that the tile kernel wins). The 256 x 256 tile kernel above would put 56
// workgroups on 256 CUs and stream the 58.7 MB of Llama-3-8B FFN weights at ~1 TB/s. Here the work is a weight stream:
// one wave per 16 weight rows (n), weights never touch LDS. Per 128-deep K block a lane loads its 32 bytes of b^T (row
// n = lane % 16, 16-byte chunks lane / 16 and lane / 16 + 4: the same k order as above) through a kD-deep register
// ring with static slots; one v_mfma_scale_f32_16x16x128_f8f6f4 per m-tile and K block, block scale on the VALU as
// above. A workgroup is 4 waves = 4 / KS n-tiles x KS waves that share an n-tile and take the K blocks kb = KS i + kpart
// (interleaved); their partial sums meet in LDS and are added in a fixed order. KS is chosen by the host so that the
// launch has ~200+ workgroups: a CU on its own pulls ~20 GB/s through its L1, so N = 4096 with KS = 1 (64 workgroups)
// ran at 1.2 TB/s. With more than 16 rows (MF >= 2) the activations of a step - [16 MF rows] x [KS K blocks] - are
// staged once per workgroup in LDS (LDSA below); with one m-tile every lane fetches its own 32 bytes.
// grid = (N / (16 * 4 / KS), ceil(M / (16 MF))); operands swapped as above: a lane owns 4 consecutive n of one m.
// MODE_FP8_ROWCOL / MODE_INT8_ROWCOL (fp8_scaled_mm / int8_scaled_mm): the same stream, the MFMAs chain into the
// accumulator and the epilogue applies sa[m] * sb[n] (+ bias) in the tile kernel's rounding order.
// LA (one m-tile only): stage the activations in LDS as the MF >= 2 forms do. With 8 .. 16 rows every wave pulls 2 KB of
// activations per K block through the L1 next to its 2 KB of weights (M = 16: 18.4 us against 13.4 at one row).
template <typename OutT, int MODE, int MF, int KS, bool HW_SCALE, bool LA = false>  // KS waves split the K blocks of one 16-row n-tile
__global__ __launch_bounds__(256) void gemm_8bit_skinny_kernel(
    OutT* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ sa, const float* __restrict__ sb, const OutT* __restrict__ bias, int M, int N, int K,
    int64_t lda, int64_t ldb, int64_t ldc, int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn) {
  constexpr bool kBlockwise = MODE == MODE_BLOCKWISE;
  using AccT = typename std::conditional<MODE == MODE_INT8_ROWCOL, v4i, v4f>::type;
  constexpr int kD = 8;  // K blocks of weights in flight per wave
  // LDSA: the activation tiles of a step ([KS blocks][16 MF rows][128 B]) are staged once per workgroup in LDS instead
  // of being fetched by each wave in MFMA layout (at MF = 4 that was 128 B of activations per lane and block against 32 B
  // of weights): global -> registers two steps ahead -> LDS one step ahead, two buffers, an LDS-only barrier per step.
  // A wave without weight rows (n0 >= N) runs along for the staging and the barriers.
  constexpr bool LDSA = MF >= 2 || LA;
  constexpr int kTile = 16 * MF * 128;                    // one K block of activations
  constexpr int kStage = LDSA ? 2 * KS * kTile : 0;       // two buffers of KS blocks
  constexpr int kRed = KS > 1 ? 4 * MF * 256 * 4 : 0;     // partial sums (after the loop: shares the staging memory)
  constexpr int kSmem = kStage > kRed ? kStage : (kRed > 16 ? kRed : 16);
  __shared__ __attribute__((aligned(1024))) char smem[kSmem];
  float* red = reinterpret_cast<float*>(smem);
  char* abuf = smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = (blockIdx.x * (4 / KS) + wave / KS) * 16;
  const int kpart = wave % KS;
  const bool live = n0 < N;  // (dead waves still join the barriers)
  const int m0 = blockIdx.y * (16 * MF);
  const int j = lane & 15, g = lane >> 4;
  const int nkb_all = K / BK;
  const int nsteps = (nkb_all + KS - 1) / KS;        // steps of the workgroup
  const int nmine = (nkb_all - kpart + KS - 1) / KS;  // K blocks of this wave: kpart, kpart + KS, ... (nsteps or one less)
  const int nkb = (live || LDSA) ? nsteps : 0;

  int nrow = n0 + j;
  nrow = nrow < N ? nrow : N - 1;
  const uint8_t* bl = b + (int64_t)nrow * ldb + g * 16 + kpart * BK;
  const uint8_t* al[MF];
  const float* sl[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    int m = m0 + mf * 16 + j;
    m = m < M ? m : M - 1;
    al[mf] = a + (int64_t)m * lda + g * 16 + kpart * BK;
    sl[mf] = kBlockwise ? sa + (int64_t)m * sa_sm + (int64_t)kpart * sa_sk : sa;
  }
  const int nblk_max = (N + 127) / 128 - 1;
  int nblk = n0 >> 7;
  nblk = nblk < nblk_max ? nblk : nblk_max;
  const float* sbw = kBlockwise ? sb + (int64_t)nblk * sb_sn + (int64_t)kpart * sb_sk : sb;

  auto load32 = [](const uint8_t* p) -> v8i {
    const v4i lo = *reinterpret_cast<const v4i*>(p), hi = *reinterpret_cast<const v4i*>(p + 64);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  AccT acc[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mf][r] = 0;
  const v4f zero = {0.f, 0.f, 0.f, 0.f};

  if (nkb > 0) {
    v8i af[2][MF];
    float sv[2][MF], sbq[2];
    // LDSA staging: 16-byte chunk cid = tid + 256 i of the [16 MF rows][KS blocks][8 chunks] step (a row's KS * 128 bytes are
    // contiguous in global memory): block tile kblk, row, chunk at (chunk ^ (row & 7))
    constexpr int AL = LDSA ? (MF * KS >= 2 ? MF * KS / 2 : 1) : 1;  // (MF = KS = 1: 128 chunks, the upper half of the threads repeat them)
    v4i areg[AL];
    const uint8_t* ag[AL];
    int aoff[AL], akb[AL];
    if constexpr (LDSA) {
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        const int cid = (threadIdx.x + 256 * i) % (128 * MF * KS), row = cid / (8 * KS), c = cid % (8 * KS), kblk = c >> 3, ch = c & 7;
        int m = m0 + row;
        m = m < M ? m : M - 1;
        ag[i] = a + (int64_t)m * lda + ch * 16;
        akb[i] = kblk;
        aoff[i] = kblk * kTile + row * 128 + ((ch ^ (row & 7)) << 4);
      }
    }
    auto load_stage = [&](int kb) {  // activations of step kb -> registers (blocks past the end: block 0, never multiplied)
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        const int kk = kb * KS + akb[i];
        areg[i] = *reinterpret_cast<const v4i*>(ag[i] + (int64_t)(kk < nkb_all ? kk : 0) * BK);
      }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
      for (int i = 0; i < AL; ++i) *reinterpret_cast<v4i*>(&abuf[buf * (KS * kTile) + aoff[i]]) = areg[i];
    };
    auto load_a = [&](int kb, int slot) {
      const int kc = kb < nmine ? kb * KS : 0;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if constexpr (!LDSA) af[slot][mf] = load32(al[mf] + (int64_t)kc * BK);
        if constexpr (kBlockwise) sv[slot][mf] = sl[mf][(int64_t)kc * sa_sk];
      }
      if constexpr (kBlockwise) sbq[slot] = sbw[(int64_t)kc * sb_sk];
    };
    if constexpr (LDSA) {
      load_stage(0);
      store_stage(0);
      load_stage(1);
    }
    load_a(0, 0);
    load_a(1, 1);
    __builtin_amdgcn_sched_barrier(0);  // (activation requests in front of the weight ring's, as in the steady state)
    v8i wq[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      wq[d] = load32(bl + (int64_t)(d < nmine ? d * KS : 0) * BK);
      __builtin_amdgcn_sched_barrier(0);
    }

    // One step: multiply from ring slot u, THEN refill it (requested before its last use the new block has to live in
    // other registers and the loop end moves the ring back into place with copies, each waiting for the load into its
    // source - i.e. for the whole ring).
    auto step = [&](int u, int kb) {
      if constexpr (LDSA) {
        // step kb is in LDS buffer kb % 2 (= u % 2: kD is even); everyone is done reading the other one
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        store_stage((u & 1) ^ 1);
        load_stage(kb + 2);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int row = mf * 16 + j;
          const char* rb = &abuf[(u & 1) * (KS * kTile) + kpart * kTile + row * 128];
          const v4i lo = *reinterpret_cast<const v4i*>(rb + ((g ^ (row & 7)) << 4));
          const v4i hi = *reinterpret_cast<const v4i*>(rb + (((g + 4) ^ (row & 7)) << 4));
          af[u & 1][mf] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
      if (KS == 1 || kb < nmine) {  // (wave-uniform; only the last step of a K that is not a multiple of KS blocks)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          if constexpr (MODE == MODE_INT8_ROWCOL) {
            acc[mf] = mfma_i8_k128(wq[u], af[u & 1][mf], acc[mf]);
          } else if constexpr (MODE == MODE_FP8_ROWCOL) {
            acc[mf] = mfma_k128<HW_SCALE>(wq[u], af[u & 1][mf], acc[mf]);
          } else {
            const v4f cur = mfma_k128<HW_SCALE>(wq[u], af[u & 1][mf], zero);
            const float sc = sv[u & 1][mf] * sbq[u & 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mf][r] = __builtin_fmaf(cur[r], sc, acc[mf][r]);
          }
        }
      }
      // (activations / scales first: vmcnt retires in order, so the wait for their 2-deep ring two steps from now also
      // waits for every weight request in front of it - this way the two youngest weight blocks stay in flight behind it)
      load_a(kb + 2, u & 1);
      const int kn = kb + kD;
      wq[u] = load32(bl + (int64_t)(kn < nmine ? kn * KS : 0) * BK);
    };
    // whole groups of kD steps without a branch (a conditional step merges its loads with the old registers through
    // copies), then the rest
    int kb0 = 0;
    for (; kb0 + kD <= nkb; kb0 += kD) {
#pragma unroll
      for (int u = 0; u < kD; ++u) step(u, kb0 + u);
    }
#pragma unroll
    for (int u = 0; u < kD; ++u)
      if (kb0 + u < nkb) step(u, kb0 + u);
  }

  if constexpr (KS > 1) {
    // the KS partial sums of an n-tile meet in LDS and are added in a fixed order by the first wave of the group
    if constexpr (LDSA) __syncthreads();  // (the staging buffers are read until the last step)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) *reinterpret_cast<AccT*>(&red[((wave * MF + mf) * 64 + lane) * 4]) = acc[mf];
    __syncthreads();
    if (kpart != 0) return;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int u = 1; u < KS; ++u) {
        const AccT o = *reinterpret_cast<const AccT*>(&red[(((wave + u) * MF + mf) * 64 + lane) * 4]);
        acc[mf][0] += o[0]; acc[mf][1] += o[1]; acc[mf][2] += o[2]; acc[mf][3] += o[3];
      }
    }
  }
  if (!live) return;

  // lane: column m = m0 + 16 mf + j, rows n = n0 + 4 g + r
  const int n = n0 + g * 4;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = m0 + mf * 16 + j;
    if (m >= M) continue;
    OutT* orow = out + (int64_t)m * ldc;
    OutT v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (kBlockwise) {
        v[r] = (OutT)acc[mf][r];
      } else {
        const int nn = (n + r < N) ? (n + r) : (N - 1);
        const float t = ((float)acc[mf][r] * sa[m]) * sb[nn];
        if constexpr (MODE == MODE_FP8_ROWCOL) {
          v[r] = (OutT)t;
          if (bias != nullptr) v[r] = (OutT)((float)v[r] + (float)bias[nn]);
        } else {
          v[r] = (bias != nullptr) ? (OutT)(t + (float)bias[nn]) : (OutT)t;
        }
      }
    }
    if (n + 3 < N && (ldc % 4 == 0) && ((uintptr_t)out % 8 == 0)) {
      Vec<OutT, 4> vv;
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] = v[r];
      store_vec<OutT, 4>(orow + n, vv);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) orow[n + r] = v[r];
    }
  }
}

// Main-loop variant: 4 = the pipelined persistent kernel. Everything else exists only in the diagnostic build
// (-DSGLK_PROBES: build.py --probes -> libsglk_probes.so, used by tools/kbench): 0, 1 = earlier main loops; 8, 9,
// 14..18 = timing probes whose RESULTS ARE GARBAGE by design. The release library has no switch and no probe code.
#ifdef SGLK_PROBES
#define SGLK_HW(hw, T, F) if (hw) { T; } else { F; }
#else
#define SGLK_HW(hw, T, F) { T; }
#endif
#ifdef SGLK_PROBES
static int g_gemm_variant = 4;
static uint32_t* g_gemm_stamps = nullptr;
#else
constexpr int g_gemm_variant = 4;
constexpr uint32_t* g_gemm_stamps = nullptr;
#endif
// which workgroups of gemm_fp8bw_x32_kernel run their half tile first (see the kernel); the diagnostic build can change it
// and can make the workgroups leave clock stamps (sglk_debug_set_gemm_stamps)
#ifdef SGLK_PROBES
static int g_skinny_la_rows = 4;
#else
constexpr int g_skinny_la_rows = 4;
#endif
#ifdef SGLK_PROBES
static int g_gemm_stagger = 0;
static uint32_t* g_clock_stamps = nullptr;
#else
constexpr int g_gemm_stagger = 0;
constexpr uint32_t* g_clock_stamps = nullptr;
#endif

// K split of fp8_blockwise_scaled_mm (round 5, late): few rows over a deep K leave the tile pipeline a handful of long units -
// M = 129 .. 512, N = 4096, K = 14336 (Llama-3-8B's down projection) is 32 .. 64 half tiles of 112 K blocks on 256 CUs, 84 - 90 us
// whichever kernel ran. With a caller-provided workspace the units are tile x K slice (S slices, fp32 partial tiles into S slabs
// of [M, N]) and a second kernel adds the slabs in slice order (deterministic) and rounds once. Returns S (0: no split).
#ifdef SGLK_PROBES
static int g_gemm_splitk = -1;  // -1: the rule below; 0: never; S: forced where the shape allows
#else
constexpr int g_gemm_splitk = -1;
#endif
static int fp8bw_splitk_slices(int64_t M, int64_t N, int64_t K) {
  if (g_gemm_splitk == 0 || g_gemm_variant != 4) return 0;
  if (M <= 48 || M > 1024 || N % 8 != 0 || K % BK != 0) return 0;
  // (lease zv, forced slice counts against the rule, one box: N = 4096, K = 14336 - 8 slices 27 - 36 us at 65 .. 256 rows against 50 - 86
  //  unsplit, 4 slices 46 / 53 at 384 / 512 against 88 / 91, 2 slices 76 / 85 at 768 / 1024 against 92 / 96; 8192^2 4 slices 28 - 37
  //  against 52 - 55 up to 256 rows; N = 6144, K = 4096 4 slices of 8 blocks 22 - 27 against 28 (65 .. 128 rows) and 55 (192, 256);
  //  N = 14336, K = 4096 2 slices 27 - 31 against 33 up to 128 rows. Up to 128 rows a small weight matrix is faster as one stream:
  //  N = K = 4096 18.5 us against 20 - 26 for every slice count.)
  //  49 .. 64 rows: N = 4096, K = 14336 and 8192^2 26 - 28 us sliced against 29 - 32; N = 6144, K = 4096 21 against 17 - 18.)
  if (g_gemm_splitk < 0 && ((M <= 128 && N * K <= (20ll << 20)) || (M <= 64 && N * K <= (32ll << 20)))) return 0;
  const int64_t units = 2 * cdiv(M, 256) * cdiv(N, 256), nkb = K / BK;  // half tiles (a 256-row tile's empty lower half runs along)
  for (int s = 8; s >= 2; s >>= 1) {
    if (g_gemm_splitk > 0 && s != g_gemm_splitk) continue;
    if (nkb % s == 0 && nkb / s >= (g_gemm_splitk > 0 ? 3 : 8) && units * s <= num_cus()) return s;
  }
  return 0;
}

template <typename OutT>
__global__ __launch_bounds__(256) void gemm_splitk_sum_kernel(OutT* __restrict__ out, const float* __restrict__ ws, int64_t M,
                                                               int64_t N, int64_t ldc, int slices) {
  // 8 columns per thread: slabs added in slice order, rounded once
  const int64_t n8 = N >> 3, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * n8) return;
  const int64_t m = i / n8, c = (i - m * n8) << 3;
  const float* p = ws + m * N + c;
  float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
  for (int s = 1; s < slices; ++s) {
    const float4 b0 = *reinterpret_cast<const float4*>(p + (int64_t)s * M * N), b1 = *reinterpret_cast<const float4*>(p + (int64_t)s * M * N + 4);
    a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
  }
  Vec<OutT, 8> v;
  v[0] = (OutT)a0.x; v[1] = (OutT)a0.y; v[2] = (OutT)a0.z; v[3] = (OutT)a0.w;
  v[4] = (OutT)a1.x; v[5] = (OutT)a1.y; v[6] = (OutT)a1.z; v[7] = (OutT)a1.w;
  store_vec<OutT, 8>(out + m * ldc + c, v);
}

// fp8_scaled_mm / int8_scaled_mm with K slices: the half-tile phase of the persistent pipeline over tile x slice units, raw
// accumulators into slabs; then the sum with the epilogue (row scale, column scale, bias in the tile kernel's rounding order).
template <int MODE, bool HW_SCALE>
__global__ __launch_bounds__(512) void gemm_8bit_slices_kernel(float* __restrict__ ws, const uint8_t* __restrict__ a,
                                                               const uint8_t* __restrict__ b, const float* __restrict__ sa,
                                                               const float* __restrict__ sb, int M, int N, int K, int64_t lda,
                                                               int64_t ldb, int tiles_m, int tiles_n, int ksplit, int64_t slab) {
  __shared__ __attribute__((aligned(256))) char smem[kStages * kStageBytes];
  gemm_8bit_persist_phase<float, MODE, HW_SCALE, 0, 4>(smem, ws, a, b, sa, sb, nullptr, M, N, K, lda, ldb, /*ldc=*/N, 0, 0, 0, 0,
                                                       tiles_m, tiles_n, 1, nullptr, nullptr, nullptr, ksplit, slab);
}

template <typename OutT, int MODE>
__global__ __launch_bounds__(256) void gemm_rowcol_slices_sum_kernel(OutT* __restrict__ out, const float* __restrict__ ws,
                                                                     const float* __restrict__ sa, const float* __restrict__ sb,
                                                                     const OutT* __restrict__ bias, int64_t M, int64_t N,
                                                                     int64_t ldc, int slices) {
  const int64_t n8 = N >> 3, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * n8) return;
  const int64_t m = i / n8, c = (i - m * n8) << 3;
  float acc[8];
  if constexpr (MODE == MODE_INT8_ROWCOL) {
    const int* p = reinterpret_cast<const int*>(ws) + m * N + c;
    int t[8];
    const v4i a0 = *reinterpret_cast<const v4i*>(p), a1 = *reinterpret_cast<const v4i*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = a0[j]; t[4 + j] = a1[j]; }
    for (int s = 1; s < slices; ++s) {
      const v4i b0 = *reinterpret_cast<const v4i*>(p + (int64_t)s * M * N), b1 = *reinterpret_cast<const v4i*>(p + (int64_t)s * M * N + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { t[j] += b0[j]; t[4 + j] += b1[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (float)t[j];
  } else {
    const float* p = ws + m * N + c;
    const v4f a0 = *reinterpret_cast<const v4f*>(p), a1 = *reinterpret_cast<const v4f*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = a0[j]; acc[4 + j] = a1[j]; }
    for (int s = 1; s < slices; ++s) {
      const v4f b0 = *reinterpret_cast<const v4f*>(p + (int64_t)s * M * N), b1 = *reinterpret_cast<const v4f*>(p + (int64_t)s * M * N + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] += b0[j]; acc[4 + j] += b1[j]; }
    }
  }
  const float rs = sa[m];
  Vec<OutT, 8> v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = (acc[j] * rs) * sb[c + j];  // (the tile kernel's / the oracle's rounding order)
    if constexpr (MODE == MODE_FP8_ROWCOL) {
      v[j] = (OutT)t;
      if (bias != nullptr) v[j] = (OutT)((float)v[j] + (float)bias[c + j]);
    } else {
      v[j] = (bias != nullptr) ? (OutT)(t + (float)bias[c + j]) : (OutT)t;
    }
  }
  store_vec<OutT, 8>(out + m * ldc + c, v);
}

// slices of the row / column scale modes (lease zw: N = 4096, K = 14336 28 us up to 128 rows on the weight stream, then 40 / 70 /
// 117 us at 129 / 257 / 513 rows - the block-scale mode's cliff before its slices)
static int rowcol_splitk_slices(int64_t M, int64_t N, int64_t K) {
  if (g_gemm_splitk == 0 || g_gemm_variant != 4) return 0;
  if (M <= 128 || M > 1024 || N % 8 != 0 || K % BK != 0) return 0;
  // (lease zx, forced slice counts on one box, fp8 / int8 alike: N = 4096, K = 14336 8 slices 33 - 37 us at 129 .. 256 rows against 41,
  //  4 slices 49 / 54 at 384 / 512 against 70, 2 slices 79 / 86 at 768 / 1024 against 117; N = 14336, K = 4096 2 slices 35 / 39 at 129 / 192
  //  against 41; a small weight matrix stays on the weight stream - N = K = 4096 15 us up to 256 rows against 23 - 26 sliced, N = 6144
  //  even)
  if (g_gemm_splitk < 0 && N * K <= (32ll << 20)) return 0;
  const int64_t units = 2 * cdiv(M, 256) * cdiv(N, 256), nkb = K / BK;
  for (int s = 8; s >= 2; s >>= 1) {
    if (g_gemm_splitk > 0 && s != g_gemm_splitk) continue;
    if (nkb % s == 0 && nkb / s >= (g_gemm_splitk > 0 ? 2 : 8) && units * s <= num_cus()) return s;
  }
  return 0;
}

template <typename OutT, int MODE>
static int launch(hipStream_t st, void* out, const void* a, const void* b, const float* sa, const float* sb,
                  const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                  int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, bool hw_scale, void* ws = nullptr,
                  int64_t ws_bytes = 0) {
  const int tiles_m = (int)cdiv(M, BM), tiles_n = (int)cdiv(N, BN);
  const unsigned grid = (unsigned)(tiles_m * tiles_n);
  const bool vec = (N % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)out % 8 == 0);
  // persistent kernel: one workgroup per CU, a multiple of 8 so that every XCD gets the same number
  const unsigned pgrid = grid < (unsigned)num_cus() ? ((grid + 7) / 8) * 8 : (unsigned)num_cus();
  // (the row-scale DMA addresses its K blocks and rows with 32-bit byte offsets inside one buffer resource)
  // (the blockwise kernel's half-tile phase looks three K blocks ahead)
  const bool persist_ok = K / BK >= (MODE == MODE_BLOCKWISE ? 3 : 2) && N % 8 == 0 && ldc % 8 == 0 && (uintptr_t)out % 16 == 0 && ldc < (1ll << 22) &&
                          ((K / BK - 1) * sa_sk + 256 * sa_sm) * 4 < (1ll << 31) &&
                          (MODE == MODE_BLOCKWISE || ((uintptr_t)sb % 16 == 0 && (uintptr_t)bias % 16 == 0));
  // does any XCD cut its last partial round into half tiles (same rule as in the kernel)?
  bool tail_halves = false;
  {
    const int slots = (int)pgrid >> 3, q = (int)grid >> 3, r8 = (int)grid & 7;
    for (int len = q; len <= q + (r8 ? 1 : 0); ++len) {
      const int left = len - (len / slots) * slots;
      tail_halves = tail_halves || (left > 0 && 2 * left <= slots);
    }
  }
  if constexpr (MODE == MODE_BLOCKWISE) {
    const int S = ws != nullptr ? fp8bw_splitk_slices(M, N, K) : 0;
    if (S > 0 && persist_ok && ws_bytes >= (int64_t)S * M * N * 4 && (uintptr_t)ws % 16 == 0) {
      const unsigned units = 2 * grid * (unsigned)S;  // half tiles x slices
      const unsigned sgrid = units < (unsigned)num_cus() ? ((units + 7) / 8) * 8 : (unsigned)num_cus();
      gemm_fp8bw_x32_kernel<float, 0><<<sgrid, 512, 0, st>>>((float*)ws, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (int)M, (int)N,
                                                            (int)K, lda, ldb, /*ldc=*/N, sa_sm, sa_sk, sb_sk, sb_sn, tiles_m, tiles_n,
                                                            1, 0, nullptr, S, M * N);
      if (int rc = check_launch("gemm_8bit(k slices)")) return rc;
      const int64_t n = M * (N >> 3);
      gemm_splitk_sum_kernel<OutT><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((OutT*)out, (const float*)ws, M, N, ldc, S);
      return check_launch("gemm_8bit(k slices: sum)");
    }
  }
  if constexpr (MODE == MODE_FP8_ROWCOL || MODE == MODE_INT8_ROWCOL) {
    const int S = ws != nullptr ? rowcol_splitk_slices(M, N, K) : 0;
    if (S > 0 && persist_ok && ws_bytes >= (int64_t)S * M * N * 4 && (uintptr_t)ws % 16 == 0 && M * N * 4 < (1ll << 32)) {
      const unsigned units = 2 * grid * (unsigned)S;
      const unsigned sgrid = units < (unsigned)num_cus() ? ((units + 7) / 8) * 8 : (unsigned)num_cus();
      SGLK_HW(hw_scale,
              (gemm_8bit_slices_kernel<MODE, true><<<sgrid, 512, 0, st>>>((float*)ws, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (int)M,
                                                                          (int)N, (int)K, lda, ldb, tiles_m, tiles_n, S, M * N)),
              (gemm_8bit_slices_kernel<MODE, false><<<sgrid, 512, 0, st>>>((float*)ws, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (int)M,
                                                                           (int)N, (int)K, lda, ldb, tiles_m, tiles_n, S, M * N)))
      if (int rc = check_launch("gemm_8bit(k slices)")) return rc;
      const int64_t n = M * (N >> 3);
      gemm_rowcol_slices_sum_kernel<OutT, MODE><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((OutT*)out, (const float*)ws, sa, sb,
                                                                                       (const OutT*)bias, M, N, ldc, S);
      return check_launch("gemm_8bit(k slices: sum)");
    }
  }
  // few rows: the weight-streaming kernel (variant 5 forces it, 7 forces it without the K split, 6 forbids it)
  {
    // (up to 512 rows also when the tile kernel would have few units: M = 256, N = 4096, K = 14336 is 32 units of 128 x
    // 256 - 134 us there, 51 us here; from ~96 units on the tile kernel wins)
    // (round 5, late - a row sweep across this boundary: the block-scale mode's weight stream pays its promotion FMAs per row
    //  set, 17 us + 0.23 us per row at N = 14336, K = 4096 against ~33 us for the tile pipeline from 65 rows on; and at N = 4096,
    //  K = 14336 its stacked 64-row workgroups took 133 - 167 us at 257 - 512 rows against ~88 us for the half tiles. The row / column
    //  scale modes keep the wider range: 22 us at 128 rows against 40 us for their tile pipeline.)
    const bool wide_n = cdiv(N, 256) >= 32;
    const bool few = MODE == MODE_BLOCKWISE
                         ? (M <= 72 || (!wide_n && (M <= 128 || (M <= 256 && cdiv(M, 128) * cdiv(N, 256) <= 80))))
                         : (M <= 128 || (M <= 512 && cdiv(M, 128) * cdiv(N, 256) <= 80));
    if ((few && g_gemm_variant == 4) || g_gemm_variant == 5 || g_gemm_variant == 7) {
#define SGLK_GO_SKINNY_(MF, KS, LA)                                                                          \
  {                                                                                                          \
    const dim3 sg((unsigned)cdiv(N, 16 * (4 / KS)), (unsigned)cdiv(M, 16 * MF));                             \
    SGLK_HW(hw_scale,                                                                                        \
      (gemm_8bit_skinny_kernel<OutT, MODE, MF, KS, true, LA><<<sg, 256, 0, st>>>(                            \
          (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K, \
          lda, ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn)),                                                       \
      (gemm_8bit_skinny_kernel<OutT, MODE, MF, KS, false, LA><<<sg, 256, 0, st>>>(                           \
          (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K, \
          lda, ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn)))                                                       \
  }
#define SGLK_GO_SKINNY(MF, KS) SGLK_GO_SKINNY_(MF, KS, false)
      // K split over the waves of a workgroup until the launch has ~200 workgroups (at N = 14336 the split measured
      // slower: 18 -> 22 us at M = 1; at N = 4096, K = 14336 the unsplit launch has 64 workgroups: 44 us)
      const bool mf8 = M > 64 && cdiv(N, 16) >= 768 && g_gemm_variant != 5;  // 128 rows per workgroup: weights read once
      const int64_t ntile = cdiv(N, 16) * (mf8 ? 1 : cdiv(M, M <= 16 ? 16 : M <= 32 ? 32 : 64));
      const int ks = (K / BK < 8 || ntile >= 768 || g_gemm_variant == 7) ? 1 : (ntile >= 384 ? 2 : 4);
#define SGLK_GO_SKINNY_KS(MF)                                                                                \
  {                                                                                                          \
    if (ks == 1) SGLK_GO_SKINNY(MF, 1) else if (ks == 2) SGLK_GO_SKINNY(MF, 2) else SGLK_GO_SKINNY(MF, 4)    \
  }
      if (M <= 16 && M >= g_skinny_la_rows) {  // (4 .. 16 rows: the activations through LDS - M = 4 17.3 -> 16.0 us, 8 18.1 -> 16.1, 16 20.7 -> 16.2; 2 rows: no difference)
        if (ks == 1) SGLK_GO_SKINNY_(1, 1, true) else if (ks == 2) SGLK_GO_SKINNY_(1, 2, true) else SGLK_GO_SKINNY_(1, 4, true)
      } else if (M <= 16) SGLK_GO_SKINNY_KS(1)
      else if (M <= 32) SGLK_GO_SKINNY_KS(2)
      else if (!mf8) SGLK_GO_SKINNY_KS(4)
      else SGLK_GO_SKINNY(8, 1)
#undef SGLK_GO_SKINNY_KS
#undef SGLK_GO_SKINNY
#undef SGLK_GO_SKINNY_
      return check_launch("gemm_8bit(skinny)");
    }
  }
  // up to 512 rows the 128-row tiling (every tile as two halves) fills more CUs: see the kernel comment
  // (M = 1024, N = 14336: 77 us against 64 us with 256-row tiles; but when the 256-row tiles cover at most half of the CUs -
  // M = 1024, N = 4096 is 64 tiles - halves double the workgroups at 0.6x the time each)
  const bool all_halves = (M <= 512 || grid <= (unsigned)num_cus() / 2) && g_gemm_variant == 4;
  const unsigned hgrid = 2 * grid < (unsigned)num_cus() ? ((2 * grid + 7) / 8) * 8 : (unsigned)num_cus();
  const int variant = (!persist_ok && g_gemm_variant != 0 && g_gemm_variant != 1) ? 1 : g_gemm_variant;
#define SGLK_GO_VAR(V, H, VAR)                                                                               \
  gemm_8bit_kernel<OutT, MODE, V, H, VAR><<<grid, 512, 0, st>>>(                                             \
      (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K,   \
      lda, ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn, tiles_m, tiles_n)
#define SGLK_GO_PIPE(V, H, P)                                                                                \
  if (all_halves) {                                                                                          \
    gemm_8bit_persist_kernel<OutT, MODE, H, P, 4><<<hgrid, 512, 0, st>>>(                                 \
        (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K, lda, \
        ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn, tiles_m, tiles_n, 1, g_gemm_stamps, nullptr, nullptr);                                     \
  } else {                                                                                                   \
    if (tail_halves && P == 0)                                                                               \
      gemm_8bit_persist2_kernel<OutT, MODE, H, 0><<<pgrid, 512, 0, st>>>(                                    \
          (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K, lda, \
          ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn, tiles_m, tiles_n, g_gemm_stamps, nullptr, nullptr);            \
    else {                                                                                                   \
    gemm_8bit_persist_kernel<OutT, MODE, H, P, 8><<<pgrid, 512, 0, st>>>(                                 \
        (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K, lda, \
        ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn, tiles_m, tiles_n, 0, g_gemm_stamps, nullptr, nullptr);                                     \
    if (tail_halves)                                                                                         \
      gemm_8bit_persist_kernel<OutT, MODE, H, P, 4><<<pgrid, 512, 0, st>>>(                               \
          (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K, lda, \
          ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn, tiles_m, tiles_n, 0, g_gemm_stamps, nullptr, nullptr);                                   \
    }                                                                                                        \
  }
// (blockwise: the 32 x 32 x 64 schedule; the 16 x 16 x 128 one stays for the row / column scale modes and as probe 22)
#define SGLK_GO_X32(P)                                                                                       \
  if constexpr (MODE == MODE_BLOCKWISE) {                                                                    \
    gemm_fp8bw_x32_kernel<OutT, P><<<all_halves ? hgrid : pgrid, 512, 0, st>>>(                              \
        (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (int)M, (int)N, (int)K, lda, ldb, ldc, sa_sm, sa_sk, \
        sb_sk, sb_sn, tiles_m, tiles_n, all_halves ? 1 : 0, tail_halves ? g_gemm_stagger : 0, g_clock_stamps, 1, 0);  \
  }
#ifdef SGLK_PROBES
#define SGLK_GO(V, H)                                                                                        \
  if constexpr (MODE == MODE_BLOCKWISE) {                                                                    \
    switch (variant) {                                                                                       \
      case 22: SGLK_GO_PIPE(V, H, 0); break;                                                                 \
      case 23: SGLK_GO_X32(1); break;                                                                        \
      case 24: SGLK_GO_X32(2); break;                                                                        \
      case 25: SGLK_GO_X32(3); break;                                                                        \
      case 26: SGLK_GO_X32(4); break;                                                                        \
      case 27: SGLK_GO_X32(5); break;                                                                        \
      case 28: SGLK_GO_X32(6); break;                                                                        \
      case 30: SGLK_GO_X32(7); break;  /* no stagger */                                                      \
      case 31: SGLK_GO_X32(8); break;                                                                        \
      case 32: SGLK_GO_X32(10); break;                                                                       \
      case 33: SGLK_GO_X32(11); break;                                                                       \
      case 34: SGLK_GO_X32(12); break;                                                                       \
      case 35: SGLK_GO_X32(13); break;                                                                       \
      case 36: SGLK_GO_X32(14); break;                                                                       \
      case 37: SGLK_GO_X32(15); break;  /* whole-tile stores staged through LDS */                             \
      case 38: SGLK_GO_X32(16); break;  /* odd units of a workgroup walk K backwards (shared a panel from L2) */    \
      case 39: SGLK_GO_X32(17); break;  /* no promotion FMAs */                                                \
      case 40: SGLK_GO_X32(18); break;  /* no LDS fragment reads */                                            \
      case 41: SGLK_GO_X32(19); break;  /* neither */                                                          \
      case 0: SGLK_GO_VAR(V, H, 0); break;                                                                   \
      case 1: SGLK_GO_VAR(V, H, 1); break;                                                                   \
      case 8: SGLK_GO_VAR(V, H, 8); break;                                                                   \
      case 9: SGLK_GO_VAR(V, H, 9); break;                                                                   \
      case 14: SGLK_GO_PIPE(V, H, 1); break;                                                                 \
      case 15: SGLK_GO_PIPE(V, H, 2); break;                                                                 \
      case 16: SGLK_GO_PIPE(V, H, 3); break;                                                                 \
      case 17: SGLK_GO_PIPE(V, H, 4); break;                                                                 \
      case 18: SGLK_GO_PIPE(V, H, 5); break;                                                                 \
      case 20: SGLK_GO_PIPE(V, H, 6); break;                                                                 \
      case 21: SGLK_GO_PIPE(V, H, 7); break;                                                                 \
      default: SGLK_GO_X32(0); break;                                                                        \
    }                                                                                                        \
  } else {                                                                                                   \
    if (variant == 0 || variant == 1) { SGLK_GO_VAR(V, H, 0); } else { SGLK_GO_PIPE(V, H, 0) }               \
  }
#else
#define SGLK_GO(V, H)                                                                                        \
  if constexpr (MODE == MODE_BLOCKWISE) {                                                                    \
    if (variant == 1) { SGLK_GO_VAR(V, H, 1); } else { SGLK_GO_X32(0) }                                      \
  } else {                                                                                                   \
    if (variant == 1) { SGLK_GO_VAR(V, H, 0); } else { SGLK_GO_PIPE(V, H, 0) }                               \
  }
#endif
  if (vec) {
    SGLK_HW(hw_scale, SGLK_GO(true, true), SGLK_GO(true, false))
  } else {
    SGLK_HW(hw_scale, SGLK_GO(false, true), SGLK_GO(false, false))
  }
#undef SGLK_GO
#undef SGLK_GO_VAR
#undef SGLK_GO_PIPE
#undef SGLK_GO_X32
  return check_launch("gemm_8bit");
}

static int check_common(const char* op, const void* a, const void* b, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldb, int out_dtype) {
  SGLK_REQUIRE(M >= 0 && N > 0 && K > 0, "%s: bad shape M=%lld N=%lld K=%lld", op, (long long)M, (long long)N,
               (long long)K);
  SGLK_REQUIRE(K % 128 == 0, "%s: K=%lld must be a multiple of 128", op, (long long)K);
  SGLK_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "%s: shape too large", op);
  SGLK_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0,
               "%s: mat_a / mat_b rows must be 16-byte aligned", op);
  SGLK_REQUIRE(lda * 256 < (1ll << 32) && ldb * 256 < (1ll << 32), "%s: leading dimension too large", op);
  SGLK_REQUIRE(out_dtype == SGLK_BF16 || out_dtype == SGLK_F16, "%s: out_dtype must be Half or BFloat16", op);
  return SGLK_OK;
}

}  // namespace

// QServe W4A8 on the persistent pipeline (called from qserve_w4a8.hip above 128 rows). Returns 0 when the shape does not
// qualify (the caller falls back to its own tile kernel), 1 after launching.
int qserve_w4a8_persist(hipStream_t st, bool group, void* out, const void* a, const void* w, const void* zeros,
                        const void* scales_i8, const void* wscales, const void* ascales, const void* w_szs,
                        const void* a_ssums, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc) {
  if (K % BK != 0 || K / BK < 2 || N % 32 != 0 || ldc % 4 != 0 || (uintptr_t)out % 8 != 0 || ldc >= (1ll << 22) ||
      lda * 256 >= (1ll << 32) || (uintptr_t)wscales % 8 != 0 || (!group && (uintptr_t)w_szs % 8 != 0) ||
      (K >> 7) * N >= (1ll << 31) || (group && ((uintptr_t)scales_i8 % 4 != 0 || (uintptr_t)zeros % 4 != 0)))
    return 0;
  const int tiles_m = (int)cdiv(M, BM), tiles_n = (int)cdiv(N, BN);
  const unsigned grid = (unsigned)(tiles_m * tiles_n);
  const unsigned pgrid = grid < (unsigned)num_cus() ? ((grid + 7) / 8) * 8 : (unsigned)num_cus();
  bool tail_halves = false;
  {
    const int slots = (int)pgrid >> 3, q = (int)grid >> 3, r8 = (int)grid & 7;
    for (int len = q; len <= q + (r8 ? 1 : 0); ++len) {
      const int left = len - (len / slots) * slots;
      tail_halves = tail_halves || (left > 0 && 2 * left <= slots);
    }
  }
  const bool all_halves = M <= 512 || grid <= (unsigned)num_cus() / 2;
  const unsigned hgrid = 2 * grid < (unsigned)num_cus() ? ((2 * grid + 7) / 8) * 8 : (unsigned)num_cus();
  // kernel arguments: sa <- ascales, sb <- wscales, bias <- w_szs, x0 <- a_ssums (per channel) / scales_i8 (per group),
  // x1 <- zeros (per group)
#define SGLK_GO_W4(MODE, MS, G, AH)                                                                          \
  gemm_8bit_persist_kernel<f16, MODE, true, 0, MS><<<G, 512, 0, st>>>(                                        \
      (f16*)out, (const uint8_t*)a, (const uint8_t*)w, (const float*)ascales, (const float*)wscales,         \
      (const f16*)w_szs, (int)M, (int)N, (int)K, lda, 0, ldc, 0, 0, 0, 0, tiles_m, tiles_n, AH, nullptr,      \
      group ? scales_i8 : a_ssums, zeros)
#define SGLK_GO_W4M(MODE)                                                                                    \
  if (all_halves) {                                                                                          \
    SGLK_GO_W4(MODE, 4, hgrid, 1);                                                                           \
  } else if (tail_halves) { /* whole tiles, then the half tiles of the last partial round: one launch (round 5) */ \
    gemm_8bit_persist2_kernel<f16, MODE, true, 0><<<pgrid, 512, 0, st>>>(                                    \
        (f16*)out, (const uint8_t*)a, (const uint8_t*)w, (const float*)ascales, (const float*)wscales,       \
        (const f16*)w_szs, (int)M, (int)N, (int)K, lda, 0, ldc, 0, 0, 0, 0, tiles_m, tiles_n, nullptr,        \
        group ? scales_i8 : a_ssums, zeros);                                                                 \
  } else {                                                                                                   \
    SGLK_GO_W4(MODE, 8, pgrid, 0);                                                                           \
  }
  if (group) {
    SGLK_GO_W4M(MODE_W4A8_GRP)
  } else {
    SGLK_GO_W4M(MODE_W4A8_CHN)
  }
#undef SGLK_GO_W4M
#undef SGLK_GO_W4
  return 1;
}

}  // namespace sglk

// The K=128 fp8 MFMA is issued in its MX encoding with unit E8M0 scales (twice the rate of the plain encoding, the
// same results: tests/test_gemm_gpu.py compares the two through the diagnostic build's hook).
#ifdef SGLK_PROBES
static int g_fp8_hw_scale = 1;
extern "C" SGLK_API void sglk_debug_set_fp8_mfma_form(int hw_scale) { g_fp8_hw_scale = hw_scale; }
extern "C" SGLK_API void sglk_debug_set_gemm_variant(int v) { sglk::g_gemm_variant = v; }
extern "C" SGLK_API void sglk_debug_set_gemm_stamps(uint32_t* p) { sglk::g_gemm_stamps = p; sglk::g_clock_stamps = p; }
extern "C" SGLK_API void sglk_debug_set_gemm_stagger(int s) { sglk::g_gemm_stagger = s; }
extern "C" SGLK_API void sglk_debug_set_skinny_la_rows(int r) { sglk::g_skinny_la_rows = r; }
extern "C" SGLK_API void sglk_debug_set_gemm_splitk(int s) { sglk::g_gemm_splitk = s; }
#else
constexpr int g_fp8_hw_scale = 1;
#endif

extern "C" int sglk_fp8_blockwise_scaled_mm(sglk_stream_t stream, void* out, const void* a, const void* b,
                                            const float* sa, const float* sb, int64_t M, int64_t N,
                                            int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                            int64_t sa_stride_m, int64_t sa_stride_k,
                                            int64_t sb_stride_k, int64_t sb_stride_n, int out_dtype) {
  using namespace sglk;
  if (int rc = check_common("fp8_blockwise_scaled_mm", a, b, M, N, K, lda, ldb, out_dtype)) return rc;
  if (M == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == SGLK_BF16)
    return launch<bf16, MODE_BLOCKWISE>(st, out, a, b, sa, sb, nullptr, M, N, K, lda, ldb, ldc, sa_stride_m,
                                        sa_stride_k, sb_stride_k, sb_stride_n, g_fp8_hw_scale != 0);
  return launch<f16, MODE_BLOCKWISE>(st, out, a, b, sa, sb, nullptr, M, N, K, lda, ldb, ldc, sa_stride_m,
                                     sa_stride_k, sb_stride_k, sb_stride_n, g_fp8_hw_scale != 0);
}

extern "C" int64_t sglk_fp8_blockwise_scaled_mm_workspace_size(int64_t M, int64_t N, int64_t K) {
  using namespace sglk;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)fp8bw_splitk_slices(M, N, K) * M * N * 4;
}

extern "C" int sglk_fp8_blockwise_scaled_mm_ws(sglk_stream_t stream, void* out, const void* a, const void* b,
                                               const float* sa, const float* sb, int64_t M, int64_t N,
                                               int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                               int64_t sa_stride_m, int64_t sa_stride_k,
                                               int64_t sb_stride_k, int64_t sb_stride_n, int out_dtype,
                                               void* workspace, int64_t workspace_bytes) {
  using namespace sglk;
  if (int rc = check_common("fp8_blockwise_scaled_mm", a, b, M, N, K, lda, ldb, out_dtype)) return rc;
  if (M == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == SGLK_BF16)
    return launch<bf16, MODE_BLOCKWISE>(st, out, a, b, sa, sb, nullptr, M, N, K, lda, ldb, ldc, sa_stride_m,
                                        sa_stride_k, sb_stride_k, sb_stride_n, g_fp8_hw_scale != 0, workspace, workspace_bytes);
  return launch<f16, MODE_BLOCKWISE>(st, out, a, b, sa, sb, nullptr, M, N, K, lda, ldb, ldc, sa_stride_m,
                                     sa_stride_k, sb_stride_k, sb_stride_n, g_fp8_hw_scale != 0, workspace, workspace_bytes);
}

extern "C" int64_t sglk_scaled_mm_workspace_size(int64_t M, int64_t N, int64_t K) {
  using namespace sglk;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)rowcol_splitk_slices(M, N, K) * M * N * 4;
}

extern "C" int sglk_scaled_mm_ws(sglk_stream_t stream, void* out, const void* a, const void* b, const float* sa,
                                 const float* sb, const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda,
                                 int64_t ldb, int64_t ldc, int in_dtype, int out_dtype, void* workspace,
                                 int64_t workspace_bytes) {
  using namespace sglk;
  const char* op = in_dtype == SGLK_INT8 ? "int8_scaled_mm" : "fp8_scaled_mm";
  SGLK_REQUIRE(in_dtype == SGLK_INT8 || in_dtype == SGLK_FP8_E4M3, "scaled_mm: inputs must be Int8 or Float8_e4m3fn");
  if (int rc = check_common(op, a, b, M, N, K, lda, ldb, out_dtype)) return rc;
  if (M == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool hw = g_fp8_hw_scale != 0;
  void* w = workspace;
  const int64_t wb = workspace_bytes;
  if (in_dtype == SGLK_INT8) {
    if (out_dtype == SGLK_BF16)
      return launch<bf16, MODE_INT8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw, w, wb);
    return launch<f16, MODE_INT8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw, w, wb);
  }
  if (out_dtype == SGLK_BF16)
    return launch<bf16, MODE_FP8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw, w, wb);
  return launch<f16, MODE_FP8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw, w, wb);
}

extern "C" int sglk_scaled_mm(sglk_stream_t stream, void* out, const void* a, const void* b, const float* sa,
                              const float* sb, const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda,
                              int64_t ldb, int64_t ldc, int in_dtype, int out_dtype) {
  using namespace sglk;
  const char* op = in_dtype == SGLK_INT8 ? "int8_scaled_mm" : "fp8_scaled_mm";
  SGLK_REQUIRE(in_dtype == SGLK_INT8 || in_dtype == SGLK_FP8_E4M3, "scaled_mm: inputs must be Int8 or Float8_e4m3fn");
  if (int rc = check_common(op, a, b, M, N, K, lda, ldb, out_dtype)) return rc;
  if (M == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool hw = g_fp8_hw_scale != 0;
  if (in_dtype == SGLK_INT8) {
    if (out_dtype == SGLK_BF16)
      return launch<bf16, MODE_INT8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw);
    return launch<f16, MODE_INT8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw);
  }
  if (out_dtype == SGLK_BF16)
    return launch<bf16, MODE_FP8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw);
  return launch<f16, MODE_FP8_ROWCOL>(st, out, a, b, sa, sb, bias, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, hw);
}
