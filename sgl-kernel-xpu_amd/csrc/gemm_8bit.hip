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
#include <type_traits>

#include "common.h"

namespace sglk {
namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 128;
constexpr int kTileBytes = BM * BK;              // 32 KiB per operand per stage
constexpr int kStageBytes = 2 * kTileBytes + 1024;  // + 256 fp32 row scales
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

enum { MODE_BLOCKWISE = 0, MODE_FP8_ROWCOL = 1, MODE_INT8_ROWCOL = 2 };

// fp8 e4m3 x e4m3, K = 128, D = A*B + C
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

template <typename OutT, int MODE, bool VEC_STORE, bool HW_SCALE>
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
  stage(0, 0);

  for (int kb = 0; kb < nkb; ++kb) {
    const int s = kb & 1;
    // the DMA of block kb has landed for every wave, and every wave has finished reading stage s^1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb + 1 < nkb) stage(kb + 1, s ^ 1);

    const char* ta = smem + s * kStageBytes;  // rows of a   -> MFMA B operand (columns = m)
    const char* tb = ta + kTileBytes;         // rows of b^T -> MFMA A operand (rows = n)

    v8i nfr[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) nfr[nf] = read_frag(tb, (wn * 64 + nf * 16) * 128 + frag_off);

    if constexpr (MODE == MODE_BLOCKWISE) {
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

template <typename OutT, int MODE>
static int launch(hipStream_t st, void* out, const void* a, const void* b, const float* sa, const float* sb,
                  const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                  int64_t sa_sm, int64_t sa_sk, int64_t sb_sk, int64_t sb_sn, bool hw_scale) {
  const int tiles_m = (int)cdiv(M, BM), tiles_n = (int)cdiv(N, BN);
  const unsigned grid = (unsigned)(tiles_m * tiles_n);
  const bool vec = (N % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)out % 8 == 0);
#define SGLK_GO(V, H)                                                                                       \
  gemm_8bit_kernel<OutT, MODE, V, H><<<grid, 512, 0, st>>>(                                                 \
      (OutT*)out, (const uint8_t*)a, (const uint8_t*)b, sa, sb, (const OutT*)bias, (int)M, (int)N, (int)K, \
      lda, ldb, ldc, sa_sm, sa_sk, sb_sk, sb_sn, tiles_m, tiles_n)
  if (vec) {
    if (hw_scale) SGLK_GO(true, true); else SGLK_GO(true, false);
  } else {
    if (hw_scale) SGLK_GO(false, true); else SGLK_GO(false, false);
  }
#undef SGLK_GO
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
}  // namespace sglk

// Test hook: 0 selects the plain (non-MX) encoding of the K=128 fp8 MFMA, 1 (default) the MX
// encoding with unit scales. Both must give identical results.
static int g_fp8_hw_scale = 1;
extern "C" SGLK_API void sglk_debug_set_fp8_mfma_form(int hw_scale) { g_fp8_hw_scale = hw_scale; }

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
