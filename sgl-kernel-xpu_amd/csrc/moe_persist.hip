// Grouped GEMM for MoE experts at PREFILL row counts on gfx950: out_e = A_e @ W_e^T (+ gated activation) for ragged expert
// row lists, 16-bit weights (moe_grouped_mm_nt_xe20) or 4-bit weights - int4 with / without zero points, mxfp4 (.._xe20_w4a16).
//
// Replaces, above ~200 rows per expert, the streaming kernels of moe_bf16.hip / moe_w4a16.hip (reference:
// src/sycl/GroupGemmXe20.cpp, src/sycl/GroupGemmW4A16Xe20.cpp:92-283 and their CuTe kernels): those give a wave its own
// weight stream and 64..128-row tiles, so at 512 rows per expert every column block re-reads the activations 224 times and
// every row block the weights - 11 GB through L2 for the Mixtral gate / up projection of 2048 tokens, matrix pipe busy 0.39.
// Here the work is a dense tile pipeline, the one of gemm_fp8bw_x32_kernel (gemm_8bit.hip):
//  * persistent workgroups (one per CU) walk 256 x 256 tiles (row block of an expert x column block), XCD by XCD, the
//    column blocks of a row block next to each other (they share its activations in the XCD's L2);
//  * per 64-deep K block the a tile [256 rows][128 B] and the b tile [256 weight rows][128 B] go global -> LDS by LDS-DMA
//    through buffer resources (rows past the expert's end / past N are out of range: zeros), two stages, ONE barrier per block
//    in front of its last m-step; the n-fragments of the next block are read behind that barrier into the registers the
//    last step's MFMAs have just consumed;
//  * 8 waves as 2 (m) x 4 (n), wave tile 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x16 (128 accumulator registers, the
//    MFMAs of a tile chain over all of K); lane (i, h) supplies row i, 16-byte chunk 2 s + h of the row for k-step s - the
//    same for both operands; chunk c of LDS row r sits at c ^ ((r >> 1) & 7): conflict-free ds_read_b128;
//  * MFMA row i of an n-fragment is weight row 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3) of its 32: a lane owns 16
//    consecutive output columns of one row (16-byte stores);
//  * gated epilogues (silu / gelu / clamped swiglu): a tile takes 128 gate columns and the 128 up columns that go with
//    them - n-fragment 0 of a wave is gate, 1 is up, the product is formed in the lane that holds both;
//  * int4: the 4-bit codes never reach the MFMA loop. Every thread fetches 16 bytes (32 codes of one weight row) of the
//    NEXT K block one block ahead, turns them into 32 values of the activation type - (code * scale) rounded once, exactly
//    what the reference's dequantisation produces (gemm_xe2.hpp:52-76) - and writes them into the b tile of the next stage:
//    the workgroup expands each code once (92 VALU per thread and block), not once per wave.
#include <type_traits>

#include "common.h"
#include "moe_tiles.h"

namespace sglk {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kBM = 256, kBN = 256, kBKB = 128;      // tile rows, weight rows, bytes of K per block (64 elements)
constexpr int kTile = kBM * kBKB;                    // 32 KiB per operand and stage
constexpr int kStage = 2 * kTile;

#define MP_LDS(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mp_rsrc(const void* p, uint32_t nrec) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(nrec), 0x00020000);
}

struct MpParams {
  void* out;
  const void* act;
  const void* w;        // 16-bit: [E][N][ldb] elements; int4: [E][N][K/2] bytes
  const void* scales;   // int4: [E][N][K/group] in the activation type; mxfp4: [E][N][K/32] E8M0 bytes
  const void* zeros;    // int4 with zero points (FMT 3): [E][N][K/group] in the activation type, codes unsigned
  const float* bias;    // [E][N] fp32 or nullptr (BIAS instantiations)
  int gshift;           // int4: log2(group)
  uint32_t* stamps;     // diagnostic build (sglk_debug_set_moe_clock_stamps): per workgroup {shader cycles, 100 MHz ticks, K blocks, MS}
  int prio47;           // s_setprio value of waves 4..7 (0..3)
  int blocks128;        // MS = 2 launch: 0 = only the <= 128-row remainders of 256-row blocks; 1 = all rows in 128-row blocks
                        // (remainders of at most 64 rows excepted: the caller's streaming kernels take them); 2 = the same
                        // blocks as 128 x 512 tiles (WIDE)
  int own_rem;          // 128-row blocks: 1 = a remainder of 1 .. 64 rows is a (partly empty) block of this launch too: no tail launch
  int64_t total_m;      // KSPL > 1: rows of the whole problem = rows of one fp32 slab of `out`
  const int32_t* rows;  // [E]
  int E, N, K, fuse;    // fuse: 0 none, 1 silu, 2 gelu (tanh), 3 relu2, 4 clamped swiglu (1, 2, 4 gated: N = gate + up rows)
  // The gpt-oss swiglu (the callers' code 5) arrives as fuse 4 with pairs = 1 (gate = weight row 2 n, up = row 2 n + 1 -
  // interleaved; bias likewise), act_kneg = -log2(e) * alpha and act_yadd = 1: min(gate, limit) * sigmoid(alpha gate) *
  // (clamp(up) + 1) is the clamped swiglu with two more constants (the other gated forms: act_kneg = -log2(e), act_yadd = 0).
  int pairs;
  float act_limit, act_kneg, act_yadd;
  int64_t ldb, stride_e;  // 16-bit weights: row stride / expert stride in elements
};

struct MpTile {
  const char* pa;  // first activation row of the block
  const char* pb;  // the expert's weights (+ the tile's first weight row)
  const char* ps;  // the expert's scales (+ the tile's first weight row)
  const char* pz;  // the expert's zero points (FMT 3)
  char* po;        // first output element of the tile
  const char* pbs; // the expert's bias (+ the tile's first weight row), BIAS only
  uint32_t nrec_a, nrec_b, nrec_o, nrec_s, nrec_bs;  // bytes in range (0: nothing); nrec_s: scales / zero points from ps / pz
  int ncols;       // valid output columns of the tile
};

// MS m-steps (32 rows per wave half) per K block: 4 = 256-row blocks; 2 = 128-row blocks. An expert's rows are cut in 256-row
// blocks; a remainder of at most 128 rows is a 128-row block of the second launch (MS = 2), a larger one a (partly empty)
// 256-row block: with ~512 +- 20 rows per expert (Mixtral, 2048 tokens) half of the experts have a remainder of ~20 rows, which
// as 256-row blocks cost a quarter more tiles.
// FMT: 0 16-bit weights, 1 int4 (two's-complement codes, no zero points), 2 mxfp4 (E8M0 scale per 32), 3 int4 with zero points
// BIAS: out += bias[e][n] (fp32, in front of the activation). The bias enters through the matrix pipe: the first MFMA of a
// tile's accumulator chain (a bf16 MFMA, whatever T) multiplies a weight-side operand whose k = 0, 1, 2 are the three bf16
// pieces of the fp32 bias of the lane's weight row (hi + mid + lo = the fp32 value: 3 x 8 significant bits) with an
// activation-side operand of three ones: the fp32 accumulator starts at exactly the bias, no registers are held across the
// tile and nothing waits in the store block (the reference's grouped GEMM adds the bias in its epilogue,
// src/sycl/GroupGemmW4A16Xe20.cpp:92-283; python/sgl_kernel/moe.py:574-587).
// WIDE (with MS = 4): a tile of 128 activation rows x 512 weight rows for 96 .. 191 rows per expert. The 128-row blocks of MS = 2
// do 1024 cycles of MFMAs per K block and barrier (3100 - 3200 shader cycles per block measured: 0.33 of the matrix pipe); here
// the ROLES of the operands are swapped instead - the wave grid is 2 (activation rows) x 4 (weight rows), a wave HOLDS its two
// activation fragments of a K block in registers (the n-fragment registers of the 256-row form) and STREAMS four weight
// fragments past them (the m-steps of the 256-row form): the same 32 MFMAs per wave, K block and barrier, the same register
// budget, the same LDS reads per MFMA as a 256 x 256 tile. Stage: a 16 KiB + b 64 KiB, two stages = all 160 KiB of LDS.
// Gated epilogues: n-fragments 0, 1 of a wave are gate columns, 2, 3 the up columns that go with them.
// KSPL = 2 (128-row blocks of a projection with few column blocks and a long K - the Mixtral down projection at 512 tokens is
// 128 tiles of 224 K blocks on 256 CUs): a unit is (tile, half of K); both halves store their fp32 accumulators - no
// activation, the bias in half 0 - into slab `half` of out = float [2][total_m][N]; the consumer (apply_shuffle_mul_sum's
// split form, moe_routing.hip) adds the two slabs - a two-term fp32 sum is the same in either order - and rounds once.
template <typename T, int FMT, int MS, bool BIAS = false, bool WIDE = false, int KSPL = 1>
__global__ __launch_bounds__(512) void moe_persist_kernel(MpParams p) {
  static_assert(!WIDE || MS == 4, "the wide tile streams four weight fragments per K block");
  static_assert(KSPL == 1 || (!WIDE && !BIAS), "the K split exists for the 128 x 256 and 256 x 256 tiles without a bias");
  constexpr int OES = KSPL > 1 ? 4 : 2;  // bytes per output element
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2 stages][a tile, b tile]
  constexpr bool W4 = FMT != 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int K = p.K, N = p.N;
  const int nkb = (K >> 6) / KSPL;  // K blocks of a unit, >= 2 (three stages: >= 3)
  // (an opaque scalar: left as an expression the compiler re-evaluates `fuse` with a branch ladder at every use inside the K loop)
  const bool gated = __builtin_amdgcn_readfirstlane((int)(p.fuse == 1 || p.fuse == 2 || p.fuse == 4)) != 0;
  // (gpt-oss swiglu: the tile's gate / up weight rows are 2 n / 2 n + 1 instead of n / N / 2 + n - a row scale of 2 on the weight
  //  side, used by the set-up code below and nothing else: the LDS slots, the fragments and the store block are those of the
  //  other gated epilogues)
  const int rmul = p.pairs ? 2 : 1;
  const int Nout = gated ? N >> 1 : N;
  constexpr int kRowsA = WIDE ? 128 : MS * 64;  // activation rows and weight rows (LDS slots) of a tile
  constexpr int kRowsB = WIDE ? 512 : 256;
  constexpr int kSW = kRowsB / 4;               // weight slots per wave column
  const int NB = gated ? (Nout + kRowsB / 2 - 1) / (kRowsB / 2) : (N + kRowsB - 1) / kRowsB;
  const int64_t a_row = (int64_t)K * 2;                                   // bytes
  const int64_t b_row = W4 ? (int64_t)(K >> 1) : p.ldb * 2;               // bytes per weight row
  const int64_t b_exp = W4 ? (int64_t)N * (K >> 1) : p.stride_e * 2;      // bytes per expert
  // LDS stage: a rows [64 MS][128 B], b rows [256][128 B]. 256-row blocks: two stages of 64 KiB. 128-row blocks: THREE stages of
  // 48 KiB - with 1024 cycles of MFMAs per K block and the block's data requested one block ahead the loop ran at the latency
  // of its LDS-DMA (3200 shader cycles per block; the same finding as the half tiles of gemm_fp8bw_x32_kernel): block g + 2's
  // b pieces (16-bit weights) go out in block g, block g + 3's a pieces behind its barrier, the 4-bit codes three blocks ahead.
  constexpr int NST = MS == 2 ? 3 : 2;
  constexpr int kOffB = kRowsA * kBKB, kStg = kOffB + kRowsB * kBKB;
  const int gshift = FMT == 2 ? 5 : p.gshift;
  const int kgroups = K >> gshift;                  // scales per weight row
  constexpr int kSB = FMT == 2 ? 1 : 2;             // bytes per scale

  // ---- the tiles: MB row blocks (all experts) x NB column blocks, column blocks fastest; XCD x owns a contiguous run
  // row blocks of an expert with r rows in THIS launch
  const bool b128 = (MS == 2 && p.blocks128 != 0) || WIDE;
  auto blocks_of = [&](int r) -> int {
    if (b128) return (r >> 7) + ((r & 127) > (p.own_rem ? 0 : 64) ? 1 : 0);
    const int full = r >> 8, tail = r & 255;
    return MS == 4 ? full + (tail > 128 ? 1 : 0) : ((tail > 0 && tail <= 128) ? 1 : 0);
  };
  int MB = 0;
  for (int c0 = 0; c0 < p.E; c0 += 64) {
    const int r = c0 + lane < p.E ? p.rows[c0 + lane] : 0;
    MB += __shfl(wave_inclusive_scan(blocks_of(r), lane), 63, 64);
  }
  MB = __builtin_amdgcn_readfirstlane(MB);
  // The two waves of a SIMD (w, w + 4) share its matrix pipe and issue ports; at equal priority the older one wins every
  // arbitration: stamps at the blocks' barriers showed waves 0..3 waiting ~1250 of a block's ~4100 cycles for waves 4..7.
  // A static priority for waves 4..7 swaps the roles exactly (they then wait 1350 cycles for waves 0..3) and leaves the
  // block time where it was; kept as a diagnostic knob, off by default. (Round 4: the halves taking turns at priority 1 inside
  // every K block - waves 4..7 in steps 0, 1, waves 0..3 in steps 2, 3 - changed nothing either: 1115 / 513 us against 1128 / 508
  // for the Mixtral gate / up and down projections at 512 rows per expert.)
#ifdef SGLK_PROBES
  uint32_t* const stamps_ = p.stamps;
  const int prio47_ = p.prio47;
#else  // (the release library has neither the stamps nor the priority knob)
  constexpr uint32_t* stamps_ = nullptr;
  constexpr int prio47_ = 0;
#endif
  if (wave >= 4) {
    if (prio47_ == 1) __builtin_amdgcn_s_setprio(1);
    else if (prio47_ == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio47_ == 3) __builtin_amdgcn_s_setprio(3);
  }
  const uint64_t st_c0 = stamps_ ? __builtin_amdgcn_s_memtime() : 0, st_r0 = stamps_ ? __builtin_amdgcn_s_memrealtime() : 0;
  const int nt = MB * NB * KSPL;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int q8 = nt >> 3, rem = nt & 7;
  const int run_first = xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8;
  const int run_len = q8 + (xcd < rem ? 1 : 0);
  const int n_units = run_len > slot ? (run_len - slot + slots - 1) / slots : 0;
  if (n_units == 0) return;

  auto describe = [&](int unit) -> MpTile {  // unit >= n_units: the null tile
    MpTile d;
    const bool live = unit < n_units;
    const int unit_id = live ? run_first + slot + unit * slots : 0;
    const int tile = unit_id / KSPL, kh = unit_id - tile * KSPL;  // (the halves of a tile are neighbours: one XCD)
    const int mblk = tile / NB, cb = tile - mblk * NB;
    // (expert, block of its rows) of row block mblk: lanes = experts, wave prefix sums (moe_tiles.h)
    int e = 0, row0 = 0, rows_e = 0, blk = 0, base_b = 0, base_r = 0;
    for (int c0 = 0; c0 < p.E; c0 += 64) {
      const int r = c0 + lane < p.E ? p.rows[c0 + lane] : 0;
      const int nb = blocks_of(r);
      const int ib = wave_inclusive_scan(nb, lane), ir = wave_inclusive_scan(r, lane);
      const bool hit = mblk >= base_b + ib - nb && mblk < base_b + ib;
      const unsigned long long m = __ballot(hit);
      if (m != 0) {
        const int src = __builtin_ctzll(m);
        e = c0 + src;
        blk = mblk - (base_b + __shfl(ib - nb, src, 64));
        row0 = base_r + __shfl(ir - r, src, 64);
        rows_e = __shfl(r, src, 64);
        break;
      }
      base_b += __shfl(ib, 63, 64);
      base_r += __shfl(ir, 63, 64);
    }
    e = __builtin_amdgcn_readfirstlane(e);
    blk = __builtin_amdgcn_readfirstlane(blk);
    rows_e = __builtin_amdgcn_readfirstlane(rows_e);
    const int first = (MS == 4 && !WIDE) ? blk * kBM : b128 ? blk * 128 : (rows_e & ~255);  // first row of the block inside its expert
    const int m0 = __builtin_amdgcn_readfirstlane(row0) + first;
    int rows_a = rows_e - first;
    rows_a = rows_a < kRowsA ? rows_a : kRowsA;
    const int c0 = gated ? cb * (kRowsB / 2) : cb * kRowsB;          // first output column (gated: = first gate row)
    const int cols = gated ? (Nout - c0 < kRowsB / 2 ? Nout - c0 : kRowsB / 2) : (N - c0 < kRowsB ? N - c0 : kRowsB);
    d.ncols = cols;
    // (K split: the unit's K range starts k0 elements into every row; k0 is a multiple of the scale group, host-checked)
    const int k0 = kh * nkb * 64;
    const uint32_t koff_a = (uint32_t)k0 * 2u, koff_b = W4 ? (uint32_t)(k0 >> 1) : (uint32_t)k0 * 2u;
    const uint32_t koff_s = W4 ? (uint32_t)(k0 >> gshift) * (uint32_t)kSB : 0u, koff_z = (uint32_t)(k0 >> gshift) * 2u;
    d.pa = (const char*)p.act + (int64_t)m0 * a_row + koff_a;
    const int w0 = c0 << p.pairs;  // the tile's first weight row
    d.pb = (const char*)p.w + (int64_t)e * b_exp + (int64_t)w0 * b_row + koff_b;
    d.ps = W4 ? (const char*)p.scales + ((int64_t)e * N + w0) * kgroups * kSB + koff_s : nullptr;
    d.pz = FMT == 3 ? (const char*)p.zeros + ((int64_t)e * N + w0) * kgroups * 2 + koff_z : nullptr;
    d.po = (char*)p.out + (((int64_t)kh * p.total_m + m0) * Nout + c0) * OES;
    d.pbs = BIAS ? (const char*)(p.bias + (int64_t)e * N + w0) : nullptr;
    d.nrec_bs = (live && BIAS) ? (uint32_t)((N - w0) * 4) : 0u;
    d.nrec_a = live ? (uint32_t)((int64_t)(rows_a - 1) * a_row + (int64_t)K * 2) - koff_a : 0u;
    // (b: the resource spans the expert's rows from the tile's first one to row N - 1: weight rows past N read zeros)
    d.nrec_b = live ? (uint32_t)((int64_t)(N - w0) * b_row) - koff_b : 0u;
    d.nrec_o = live ? (uint32_t)(((int64_t)(rows_a - 1) * Nout + cols) * OES) : 0u;
    // (weight rows past N of an edge tile: their scales lie past the tensor too - out of range, read as zero)
    // (FMT 3 reads scales and zero points - both two bytes per group - under this one range)
    d.nrec_s = (live && W4) ? (uint32_t)((int64_t)(N - w0) * kgroups * kSB) - koff_s : 0u;
    return d;
  };
  auto pick = [](bool c, const MpTile& x, const MpTile& y) -> MpTile {
    MpTile d;
    d.pa = c ? x.pa : y.pa;  d.pb = c ? x.pb : y.pb;  d.ps = c ? x.ps : y.ps;  d.pz = c ? x.pz : y.pz;  d.po = c ? x.po : y.po;
    d.nrec_a = c ? x.nrec_a : y.nrec_a;  d.nrec_b = c ? x.nrec_b : y.nrec_b;  d.nrec_o = c ? x.nrec_o : y.nrec_o;
    d.nrec_s = c ? x.nrec_s : y.nrec_s;
    d.pbs = c ? x.pbs : y.pbs;  d.nrec_bs = c ? x.nrec_bs : y.nrec_bs;
    d.ncols = c ? x.ncols : y.ncols;
    return d;
  };
  // weight row (relative to the tile's first) of LDS row `slot`: the wave that owns columns wn * 64 .. reads slots wn * 64 ..;
  // gated: its first 32 slots are gate rows c0 + 32 wn .., its last 32 the up rows N/2 further on
  // (wide tile: 128 slots per wave, its first 64 gate rows c0 + 64 wn .., its last 64 the up rows)
  auto wrow_of = [&](int s) -> int {
    const int gi = (s / kSW) * (kSW / 2) + (s & (kSW / 2 - 1));  // gate index of the slot inside the tile
    return !gated ? s : rmul == 2 ? 2 * gi + ((s & (kSW / 2)) ? 1 : 0) : gi + ((s & (kSW / 2)) ? Nout : 0);
  };

  const uint32_t lds_base = (uint32_t)(uintptr_t)MP_LDS(smem);
  // DMA piece = 8 LDS rows x 128 B; lane -> row lane / 8, chunk (lane % 8) ^ key(row), key = (row >> 1) & 7
  uint32_t voff_a[2], voff_b[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const uint32_t ch = (uint32_t)(((lane & 7) ^ ((par * 4 + (lane >> 4)) & 7)) << 4);
    voff_a[par] = (uint32_t)(lane >> 3) * (uint32_t)a_row + ch;
    voff_b[par] = (uint32_t)(lane >> 3) * (uint32_t)(b_row * rmul) + ch;  // (gpt-oss: a piece's 8 slots are every other weight row)
  }
  // one 1-KiB piece per call: part 0, 1 = rows of a, part 2, 3 = weight rows (16-bit weights only), sub 0, 1 each.
  // The piece's row offset inside its tile is a per-wave constant (scalar registers, set once).
  // (128-row blocks: the a tile has 16 pieces, two per wave - part 0 only, piece 2 wave + sub)
  // (wide tile: 16 a pieces as well; 64 b pieces, eight per wave: parts 2 .. 5)
  constexpr bool kA2 = MS == 2 || WIDE;
  constexpr int kBP = kRowsB / 64;  // b pieces per wave
  int a_poff[4], b_poff[8];  // (b_poff[kBP]: a constant index past a 4-entry array in the never-taken wide calls made the host pass drop the 16-bit instantiations without a diagnostic)
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
    a_poff[ii] = __builtin_amdgcn_readfirstlane((kA2 ? wave * 2 + (ii & 1) : wave * 4 + ii) * 8 * (int)a_row);
#pragma unroll
  for (int ii = 0; ii < kBP; ++ii)
    b_poff[ii] = __builtin_amdgcn_readfirstlane(wrow_of((wave * kBP + ii) * 8) * (int)b_row);  // (8 slots of a piece = 8 consecutive rows)
  auto dma_piece = [&](const MpTile& d, int kb, int s, int part, int sub) {
    char* base = smem + s * kStg;
    const int ii = part < 2 ? (part & 1) * 2 + sub : (part - 2) * 2 + sub, piece = part < 2 ? wave * 4 + ii : wave * kBP + ii;
    if (part < 2) {
      if constexpr (kA2) {
        if (part == 1) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(mp_rsrc(d.pa, d.nrec_a), MP_LDS(base + (wave * 2 + sub) * 1024), 16,
                                                 voff_a[sub], kb * kBKB + a_poff[sub], 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(mp_rsrc(d.pa, d.nrec_a), MP_LDS(base + piece * 1024), 16, voff_a[ii & 1],
                                                 kb * kBKB + a_poff[ii], 0, 0);
      }
    } else if constexpr (!W4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(mp_rsrc(d.pb, d.nrec_b), MP_LDS(base + kOffB + piece * 1024), 16,
                                               voff_b[ii & 1], kb * kBKB + b_poff[ii], 0, 0);
    }
  };

  // ---- int4 / mxfp4: thread (slot, half) expands codes [32 half, 32 half + 32) of its weight row per K block
  // (tid -> (slot, half) so that the eight lanes of a ds_write_b128 group hold eight different swizzle keys: 16-slot block
  // tid / 32, half (tid / 16) % 2, slot parity (tid / 8) % 2, key tid % 8. With slot = tid / 2 two lanes of a group shared a
  // key and a 128-byte bank window: SQ_LDS_BANK_CONFLICT was 20 % of the LDS cycles in the first r03 profile.)
  // Wide tile: 512 slots, every thread takes (slot, half) and (slot + 256, half) - the second item's registers are the "2" ones.
  const int pslot = (tid >> 5) * 16 + 2 * (tid & 7) + ((tid >> 3) & 1), phalf = (tid >> 4) & 1;
  const uint32_t pvoff_w = (uint32_t)wrow_of(pslot) * (uint32_t)b_row + (uint32_t)phalf * 16u;
  // (slot + 256 is weight row + 256 - gated: + 128, the same half of the next-but-one wave column: a scalar byte offset)
  const int prow2 = gated ? 128 * rmul : 256;
  // (the thread's 32 codes lie in ONE scale group: group (64 kb + 32 half) >> gshift of the row)
  const uint32_t pvoff_s = (uint32_t)wrow_of(pslot) * (uint32_t)kgroups * (uint32_t)kSB +
                           (uint32_t)((phalf * 32) >> gshift) * (uint32_t)kSB;
  const uint32_t pwr = (uint32_t)(pslot * 128);           // LDS row of the b tile (second item: + 256 rows)
  const uint32_t pkey = (uint32_t)((pslot >> 1) & 7);
  v4i raw_c = {0, 0, 0, 0}, raw_n = {0, 0, 0, 0}, raw_nn = {0, 0, 0, 0};  // codes of blocks + 1, + 2 (and + 3: three stages)
  uint32_t sraw_c = 0, sraw_n = 0, sraw_nn = 0;
  v4i raw_c2 = {0, 0, 0, 0};  // (wide tile: the second item; both items are re-loaded in place right behind their expansion)
  uint32_t sraw_c2 = 0;
  auto load_raw_at = [&](const MpTile& d, int kb, v4i& raw, uint32_t& sr, int drow) {
    if constexpr (W4) {
      raw = __builtin_amdgcn_raw_buffer_load_b128(mp_rsrc(d.pb, d.nrec_b), (int)pvoff_w,
                                                  kb * 32 + __builtin_amdgcn_readfirstlane(drow * (int)b_row), 0);
      const int so = ((kb * 64) >> gshift) * kSB + __builtin_amdgcn_readfirstlane(drow * kgroups * kSB);
      if constexpr (FMT == 2) {
        sr = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(mp_rsrc(d.ps, d.nrec_s), (int)pvoff_s,
                                                                     so, 0);
      } else {
        sr = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(mp_rsrc(d.ps, d.nrec_s),
                                                                       (int)pvoff_s, so, 0);
        if constexpr (FMT == 3)  // (zero point in the high half)
          sr |= (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(mp_rsrc(d.pz, d.nrec_s),
                                                                          (int)pvoff_s, so, 0) << 16;
      }
    }
  };
  auto load_raw = [&](const MpTile& d, int kb, v4i& raw, uint32_t& sr) { load_raw_at(d, kb, raw, sr, 0); };
  auto load_raw2 = [&](const MpTile& d, int kb, v4i& raw, uint32_t& sr) {
    if constexpr (WIDE) load_raw_at(d, kb, raw, sr, prow2);
  };
  // dword q of the 16 bytes -> 8 values -> chunk 4 half + q of the thread's row of the b tile at wb (stored at once: keeping the
  // four results of a block in registers until the block's barrier cost 12 registers and, with the staggered schedule, spills)
  auto expand = [&](const v4i& raw, uint32_t sr, int q, uint32_t wb) {
    v4i wx;
    auto put = [&]() {
      const uint32_t ad = wb + ((((uint32_t)(4 * phalf + q)) ^ pkey) << 4);
      asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(wx) : "memory");
    };
    if constexpr (FMT == 2) {
      // v_cvt_scalef32_pk_bf16_fp4: the two e2m1 codes of byte b (low nibble first) times 2^(E8M0 - 127), exact (moe_w4a16.hip)
      typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
      const float sc = __uint_as_float(sr ? sr << 23 : 0x00400000u);
      const uint32_t wd = (uint32_t)raw[q];
      wx[0] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wd, sc, 0));
      wx[1] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wd, sc, 1));
      wx[2] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wd, sc, 2));
      wx[3] = __builtin_bit_cast(int, (v2bf_)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wd, sc, 3));
      put();
      return;
    }
    if constexpr (FMT == 3) {  // unsigned codes: (code - zero) * scale, rounded once
      const float sc = (float)__builtin_bit_cast(T, (uint16_t)sr), zp = (float)__builtin_bit_cast(T, (uint16_t)(sr >> 16));
      const uint32_t wd = (uint32_t)raw[q];
      const uint32_t x = wd & 0x0F0F0F0Fu, y = (wd >> 4) & 0x0F0F0F0Fu;  // even / odd k
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float f0 = ((float)(uint8_t)(x >> (8 * b)) - zp) * sc, f1 = ((float)(uint8_t)(y >> (8 * b)) - zp) * sc;
        const T t0 = (T)f0, t1 = (T)f1;
        wx[b] = (int)((uint32_t)__builtin_bit_cast(uint16_t, t0) | ((uint32_t)__builtin_bit_cast(uint16_t, t1) << 16));
      }
      put();
      return;
    }
    const float s16 = (float)__builtin_bit_cast(T, (uint16_t)sr) * 0.0625f;
    const uint32_t wd = (uint32_t)raw[q];
    const uint32_t x = (wd << 4) & 0xF0F0F0F0u, y = wd & 0xF0F0F0F0u;  // signed bytes 16 * code: even / odd k
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float f0 = (float)(int8_t)(x >> (8 * b)) * s16, f1 = (float)(int8_t)(y >> (8 * b)) * s16;
      const T t0 = (T)f0, t1 = (T)f1;
      wx[b] = (int)((uint32_t)__builtin_bit_cast(uint16_t, t0) | ((uint32_t)__builtin_bit_cast(uint16_t, t1) << 16));
    }
    put();
  };

  // ---- fragment addressing (byte offsets inside a stage): lane (i, h): row i, chunk (2 s + h) ^ key(row) = (chunk h) ^ (s << 5)
  const int li = lane & 31, lh = lane >> 5;
  const int brow = ((li >> 2) & 1) * 16 + (li >> 3) * 4 + (li & 3);
  const int frag_off_a = li * 128 + ((lh ^ ((li >> 1) & 7)) << 4);
  const int frag_off_b = brow * 128 + ((lh ^ ((brow >> 1) & 7)) << 4);

  // ---- stores: lane (i, h) owns row wm * 128 + 32 mf + i; plain: columns wn * 64 + 32 nf + 16 h .. + 15 of the tile;
  // gated: columns wn * 32 + 16 h .. + 15 (gate = n-fragment 0, up = n-fragment 1)
  // (wide tile: row wm * 64 + 32 hf + i of held fragment hf; columns wn * 128 + 32 nf + 16 h .. of streamed fragment nf,
  // gated wn * 64 + 32 nf + 16 h .. for nf = 0, 1)
  const uint32_t orow_off = WIDE ? (uint32_t)(((int64_t)(wm * 64 + li) * Nout + (gated ? wn * 64 : wn * 128) + lh * 16) * 2)
                                 : (uint32_t)(((int64_t)(wm * (MS * 32) + li) * Nout + (gated ? wn * 32 : wn * 64) + lh * 16) * OES);
  auto act_mul = [&](float x, float y) -> float {
    if (p.fuse == 4) {
      x = fminf(x, p.act_limit);
      y = fminf(fmaxf(y, -p.act_limit), p.act_limit) + p.act_yadd;
    }
    // (hardware exp2 / reciprocal, ~1 ulp each: the libm forms are ~40 instructions per element, and the 64 elements per
    // lane and m-step of the store block - straight-line code - then thrash the instruction cache once per tile)
    float a;
    if (p.fuse == 1 || p.fuse == 4) {
      a = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p.act_kneg * x));  // x * sigmoid(k x), k = 1 or alpha
    } else {
      const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      // 0.5 (1 + tanh(u)) = 1 / (1 + exp(-2 u))
      a = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.885390081777927f * inner));
    }
    return a * y;
  };
  auto store_frag = [&](const MpTile& d, const v16f (&accm)[2], int mf) {
    const __amdgpu_buffer_rsrc_t ro = mp_rsrc(d.po, d.nrec_o);
    const int soff = __builtin_amdgcn_readfirstlane(mf * 32 * Nout * OES);
    if constexpr (KSPL > 1) {  // the fp32 accumulators as they are: 64 bytes per lane and n-fragment
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {  // (N is a multiple of 8: one range decision - one offset register - per eight columns)
          const int col = wn * 64 + nf * 32 + lh * 16 + hv * 8;
          const uint32_t vo = col < d.ncols ? orow_off + (uint32_t)((nf * 32 + hv * 8) * 4) : 0x80000000u;
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            // (through float temporaries: __builtin_bit_cast on a vector-element lvalue read element 0 for every index)
            const int j = hv * 8 + q2 * 4;
            const float f0 = accm[nf][j], f1 = accm[nf][j + 1], f2 = accm[nf][j + 2], f3 = accm[nf][j + 3];
            const v4i data = {(int)__float_as_uint(f0), (int)__float_as_uint(f1), (int)__float_as_uint(f2), (int)__float_as_uint(f3)};
            __builtin_amdgcn_raw_buffer_store_b128(data, ro, (int)(vo + (uint32_t)(q2 * 16)), soff, 0);
            asm volatile("s_nop 4" ::"v"(data));
          }
        }
      }
      return;
    }
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      if (gated && nf == 1) break;
#pragma unroll
      for (int hv = 0; hv < 2; ++hv) {
        Vec<T, 8> v;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float t = accm[nf][hv * 8 + c];
          if (gated) {
            t = act_mul(t, accm[1][hv * 8 + c]);
          } else if (p.fuse == 3) {
            t = fmaxf(t, 0.f);
            t = t * t;
          }
          v[c] = (T)t;
        }
        const int col = (gated ? wn * 32 : wn * 64 + nf * 32) + lh * 16 + hv * 8;
        const uint32_t vo = col < d.ncols ? orow_off + (uint32_t)((nf * 32 + hv * 8) * 2) : 0x80000000u;
        const v4i data = __builtin_bit_cast(v4i, v);
        __builtin_amdgcn_raw_buffer_store_b128(data, ro, (int)vo, soff, 0);
        asm volatile("s_nop 4" ::"v"(data));  // (store data is read for a few cycles after issue: gemm_8bit.hip)
      }
    }
  };

  // wide tile, step nf of a tile's first K block: the finished tile's columns of streamed fragment nf, both held fragments
  // (gated: nf = 0, 1 with the up values of fragment nf + 2)
  auto store_frag_w = [&](const MpTile& d, const v16f (&accm)[2], const v16f (&upm)[2], int nf) {
    const __amdgpu_buffer_rsrc_t ro = mp_rsrc(d.po, d.nrec_o);
    if (gated && nf >= 2) return;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int soff = __builtin_amdgcn_readfirstlane(hf * 32 * Nout * 2);
#pragma unroll
      for (int hv = 0; hv < 2; ++hv) {
        Vec<T, 8> v;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float t = accm[hf][hv * 8 + c];
          if (gated) {
            t = act_mul(t, upm[hf][hv * 8 + c]);
          } else if (p.fuse == 3) {
            t = fmaxf(t, 0.f);
            t = t * t;
          }
          v[c] = (T)t;
        }
        const int col = (gated ? wn * 64 : wn * 128) + nf * 32 + lh * 16 + hv * 8;
        const uint32_t vo = col < d.ncols ? orow_off + (uint32_t)((nf * 32 + hv * 8) * 2) : 0x80000000u;
        const v4i data = __builtin_bit_cast(v4i, v);
        __builtin_amdgcn_raw_buffer_store_b128(data, ro, (int)vo, soff, 0);
        asm volatile("s_nop 4" ::"v"(data));
      }
    }
  };

  v16f acc[MS][2];  // [streamed step][held fragment]: 256-row form [m-step][n-fragment], wide tile [weight fragment][row fragment]
#pragma unroll
  for (int mf = 0; mf < MS; ++mf)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

#define MP_RD16(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define MP_WR16(addr, src) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(src) : "memory")
// (the weight fragment is the MFMA's first operand either way: held nq in the 256-row form, streamed mq in the wide tile)
#define MP_MFMA(mf, nf, s)                                                                                     \
  if constexpr (std::is_same<T, bf16>::value) {                                                                \
    if constexpr (WIDE) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[mf][nf]) : "v"(mq[s]), "v"(nq[nf][s])); \
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[mf][nf]) : "v"(nq[nf][s]), "v"(mq[s])); \
  } else {                                                                                                     \
    if constexpr (WIDE) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[mf][nf]) : "v"(mq[s]), "v"(nq[nf][s])); \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[mf][nf]) : "v"(nq[nf][s]), "v"(mq[s])); \
  }

  v4i nq[2][4];  // [n-fragment][k-step]
  v4i mq[4];     // [k-step] of the running m-fragment (each re-read right behind its second MFMA)
  int gblk = 0, stg = 0;  // K blocks done; (three stages) the stage of the running block
  // BIAS: the weight-side operands of the two n-fragments (lane (i, 0): {hi, mid, lo, 0 ..} of weight row brow(i); lanes (i, 1):
  // zeros) and the activation-side operand of ones, rebuilt at the top of every tile's first K block
  // (only the two non-zero dwords of each operand live across the store block; the rest is rebuilt in front of the MFMA)
  constexpr int kNFB = WIDE ? 4 : 2;  // weight fragments of a wave
  float bq[kNFB];  // (the fp32 values live across the store block; the three pieces are made in front of each MFMA)
#pragma unroll
  for (int f = 0; f < kNFB; ++f) bq[f] = 0.f;
  const int lh_mask = lh == 0 ? -1 : 0;
  auto load_bias = [&](const MpTile& d) {
    if constexpr (BIAS) {
      const __amdgpu_buffer_rsrc_t rb = mp_rsrc(d.pbs, d.nrec_bs);
#pragma unroll
      for (int f = 0; f < kNFB; ++f) {
        bq[f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, wrow_of(wn * kSW + f * 32 + brow) * 4, 0, 0));
      }
    }
  };
  auto bias_operand = [&](float bv) -> v4i {
      // (always bf16 pieces and the bf16 MFMA, whatever T: bf16 has fp32's exponent range, so hi + mid + lo IS the fp32 value;
      // fp16 pieces of a small bias fall on the subnormal grid and came out an fp16 ulp off in 9 % of the elements)
      auto split = [&](float b) -> v4i {
        const bf16 hi = (bf16)b;
        const float r1 = b - (float)hi;
        const bf16 mid = (bf16)r1;
        const float r2 = r1 - (float)mid;
        const bf16 lo = (bf16)r2;
        return (v4i){(int)((uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, mid) << 16)) & lh_mask,
                     (int)(uint32_t)__builtin_bit_cast(uint16_t, lo) & lh_mask, 0, 0};
      };
      return split(bv);
  };
#define MP_MFMA_BIAS(mf, nf)                                                                                   \
  {                                                                                                            \
    const v4i bop_ = bias_operand(bq[WIDE ? (mf) : (nf)]);                                                     \
    const v4i one_ = {0x3F803F80 & lh_mask, 0x00003F80 & lh_mask, 0, 0};                                       \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[mf][nf]) : "v"(bop_), "v"(one_)); \
  }
#ifdef SGLK_PROBES
  constexpr bool kBarStamps = FMT <= 1;  // (the per-wave barrier stamps: diagnostic build, 16-bit and plain int4 weights)
#else
  constexpr bool kBarStamps = false;
#endif
  uint32_t st_own = 0, st_bar = 0;

  // m-step mf of a K block: k-steps s = 0..3, two MFMAs each. LAST: behind the block's barrier; its gaps carry the reads of the
  // next block's fragments (per k-step: n0, n1, m). WR: the step that ends with the int4 producer's 4 LDS stores (the step
  // in front of the barrier). The lgkmcnt of the wait in front of k-step s: step 0 9 / 7 / 5 / 3 (12 reads of the last
  // step, one re-read per k-step behind them), other steps 3.
#define MP_STEP(mf, STORE, LAST, WR)                                                                           \
  {                                                                                                            \
    if constexpr (STORE) { /* first K block of a tile: the finished tile's rows of this step leave before the step's MFMAs */ \
      if constexpr (WIDE) store_frag_w(prv, acc[mf], acc[(mf) < 2 ? (mf) + 2 : (mf)], (mf));                   \
      else store_frag(prv, acc[mf], (mf));                                                                     \
      _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f; \
      asm volatile("" : "+v"(acc[mf][0]), "+v"(acc[mf][1]));                                                   \
      if constexpr (BIAS) { /* the accumulator chain of the new tile starts at its bias */                     \
        MP_MFMA_BIAS(mf, 0) /* (s_nop in front: VALU writes of the operands / SrcC -> MFMA) */               \
        MP_MFMA_BIAS(mf, 1)                                                                                    \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                         \
      if (!(LAST)) {                                                                                           \
        if ((mf) == 0) {                                                                                       \
          if (s_ == 0) asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(nq[0][0]), "+v"(nq[1][0]), "+v"(mq[0]));     \
          if (s_ == 1) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(nq[0][1]), "+v"(nq[1][1]), "+v"(mq[1]));     \
          if (s_ == 2) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(nq[0][2]), "+v"(nq[1][2]), "+v"(mq[2]));     \
          if (s_ == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(nq[0][3]), "+v"(nq[1][3]), "+v"(mq[3]));     \
        } else {                                                                                               \
          asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(mq[s_]));                                                 \
        }                                                                                                      \
      }                                                                                                        \
      MP_MFMA(mf, 0, s_)                                                                                       \
      MP_MFMA(mf, 1, s_)                                                                                       \
      if (LAST) {                                                                                              \
        const uint32_t x_ = (uint32_t)(s_ << 5);                                                               \
        MP_RD16(nq[0][s_], nb0 ^ x_, 0);                                                                       \
        MP_RD16(nq[1][s_], nb0 ^ x_, 4096);                                                                    \
        MP_RD16(mq[s_], na0 ^ x_, 0);                                                                          \
        /* (all of a's pieces of block + 2 go out here, a whole block ahead of their barrier: with parts 1..3 in the */ \
        /* next block's first two steps the wait at the barrier cost ~2100 cycles per block)                      */ \
        if (!kA2 || s_ < 2) dma_piece(dL, kbL, sL, kA2 ? 0 : (s_ >> 1), kA2 ? s_ : (s_ & 1));                  \
      } else {                                                                                                 \
        MP_RD16(mq[s_], a0 ^ (uint32_t)(s_ << 5), ((mf) + 1) * 4096);                                          \
        if ((mf) == 0) dma_piece(dE, kbE, sE, 2 + (s_ >> 1), s_ & 1); /* (16-bit weights: the b pieces) */       \
        if (WIDE && (mf) == 1) dma_piece(dE, kbE, sE, 4 + (s_ >> 1), s_ & 1);                                  \
      }                                                                                                        \
      /* the two waves of a SIMD (w, w + 4) run this stream in lockstep: with the expansion in the same gaps of both, its   */ \
      /* VALU time adds to the block (neither has MFMAs ready for the other's VALU phase). Waves 0..3 expand in step 0,    */ \
      /* waves 4..7 in step 2: each one's VALU phase lies beside the other's bare MFMAs.                                   */ \
      if (W4 && !(LAST) && MS == 4 && !WIDE && (((mf) == 0 && wave < 4) || ((mf) == 2 && wave >= 4)))                     \
        expand(raw_c, sraw_c, s_, nbase + (uint32_t)kOffB + pwr);                                              \
      /* wide tile, two items per thread: waves 0..3 in steps 0 and 1, waves 4..7 in steps 1 and 2 */          \
      /* (each item's codes of block + 2 are requested into the same registers right behind its last expansion) */ \
      if (W4 && !(LAST) && WIDE && (mf) == (wave < 4 ? 0 : 1)) {                                               \
        expand(raw_c, sraw_c, s_, nbase + (uint32_t)kOffB + pwr);                                              \
        if (s_ == 3) load_raw(d2, kb2, raw_c, sraw_c);                                                         \
      }                                                                                                        \
      if (W4 && !(LAST) && WIDE && (mf) == (wave < 4 ? 1 : 2)) {                                               \
        expand(raw_c2, sraw_c2, s_, nbase + (uint32_t)(kOffB + 256 * 128) + pwr);                              \
        if (s_ == 3) load_raw2(d2, kb2, raw_c2, sraw_c2);                                                      \
      }                                                                                                        \
      if (W4 && !(LAST) && MS == 2) expand(raw_c, sraw_c, s_, nbase + (uint32_t)kOffB + pwr);                  \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
#define MP_BLOCK(STORE)                                                                                        \
  {                                                                                                            \
    const int s = NST == 2 ? (gblk & 1) : stg;                                                                 \
    const int s1 = NST == 2 ? (s ^ 1) : (s == 2 ? 0 : s + 1), s2 = NST == 2 ? s : (s1 == 2 ? 0 : s1 + 1);      \
    const uint32_t sbase = lds_base + (uint32_t)(s * kStg);                                                    \
    const uint32_t nbase = lds_base + (uint32_t)(s1 * kStg);                                                   \
    const bool in1 = kb + 1 < nkb, in2 = kb + 2 < nkb, in3 = kb + 3 < nkb;                                     \
    const MpTile d1 = pick(in1, cur_t, nxt), d2 = pick(in2, cur_t, nxt);                                       \
    const MpTile d3 = NST == 2 ? d2 : pick(in3, cur_t, nxt);                                                   \
    const int kb1 = in1 ? kb + 1 : 0, kb2 = in2 ? kb + 2 : kb + 2 - nkb, kb3 = in3 ? kb + 3 : kb + 3 - nkb;    \
    /* dE / kbE / sE: the b pieces issued in front of the barrier (block + 1; three stages: block + 2); dL / kbL / sL: the a */ \
    /* pieces issued behind it, into the stage just read (block + 2; three stages: block + 3)                                */ \
    const MpTile& dE = NST == 2 ? d1 : d2;                                                                     \
    const MpTile& dL = NST == 2 ? d2 : d3;                                                                     \
    const int kbE = NST == 2 ? kb1 : kb2, kbL = NST == 2 ? kb2 : kb3, sE = NST == 2 ? s1 : s2, sL = s;         \
    (void)d1; (void)kb1;                                                                                       \
    uint32_t a0, nb0 = 0, na0 = 0;                                                                             \
    {                                                                                                          \
      int fo = frag_off_a;                                                                                     \
      asm volatile("" : "+v"(fo));                                                                             \
      a0 = sbase + (uint32_t)(wm * (MS * 32) * 128) + (uint32_t)fo;                                            \
      if constexpr (WIDE) { /* the streamed fragments are the wave's 128 weight slots */                       \
        int fb = frag_off_b;                                                                                   \
        asm volatile("" : "+v"(fb));                                                                           \
        a0 = sbase + (uint32_t)(kOffB + wn * 128 * 128) + (uint32_t)fb;                                        \
      }                                                                                                        \
    }                                                                                                          \
    if constexpr (BIAS && (STORE)) load_bias(cur_t);                                                          \
    if constexpr (WIDE) { /* (see the steps) */ }                                                              \
    else if constexpr (NST == 2) load_raw(d2, kb2, raw_n, sraw_n); else load_raw(d3, kb3, raw_nn, sraw_nn);    \
    if constexpr (MS == 4) { MP_STEP(0, STORE, false, false) MP_STEP(1, STORE, false, false) MP_STEP(2, STORE, false, true) } \
    else { MP_STEP(0, STORE, false, true) }                                                                    \
    if constexpr (kBarStamps) { /* diagnostic build: where the block's barrier time goes (own data / the other waves) */ \
      if (stamps_ != nullptr) {                                                                               \
        const uint64_t t0_ = __builtin_amdgcn_s_memtime();                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(mq[0]), "+v"(mq[1]), "+v"(mq[2]), "+v"(mq[3]) : : "memory"); \
        const uint64_t t1_ = __builtin_amdgcn_s_memtime();                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                       \
        const uint64_t t2_ = __builtin_amdgcn_s_memtime();                                                     \
        st_own += (uint32_t)(t1_ - t0_);                                                                       \
        st_bar += (uint32_t)(t2_ - t1_);                                                                       \
      } else {                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(mq[0]), "+v"(mq[1]), "+v"(mq[2]), "+v"(mq[3]) : : "memory"); \
      }                                                                                                        \
    } else if constexpr (NST == 3) {                                                                           \
      /* vmcnt retires in issue order; younger than block + 1's data are this wave's 2 a pieces of block + 2 (issued behind the */ \
      /* last barrier) and, in this block, the 4 b pieces of block + 2 (16-bit weights) or the 2 - 3 code / scale loads of       */ \
      /* block + 3; the stores of a tile's first block are younger still: the count stays (it then waits for a few of them too) */ \
      constexpr int kYoung = FMT == 0 ? 6 : FMT == 3 ? 5 : 4;                                                  \
      asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)\n\ts_barrier" : "+v"(mq[0]), "+v"(mq[1]), "+v"(mq[2]), "+v"(mq[3]) : "n"(kYoung) : "memory"); \
    } else if constexpr (WIDE && W4) {                                                                         \
      /* (the block's a pieces are older than the 2 x (2 .. 3) code / scale loads this wave issued behind its expansions - waves */ \
      /* 4..7 right in front of this barrier: those may stay in flight)                                                          */ \
      constexpr int kYoungW = FMT == 3 ? 6 : 4;                                                                \
      asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)\n\ts_barrier" : "+v"(mq[0]), "+v"(mq[1]), "+v"(mq[2]), "+v"(mq[3]) : "n"(kYoungW) : "memory"); \
    } else                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(mq[0]), "+v"(mq[1]), "+v"(mq[2]), "+v"(mq[3]) : : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    {                                                                                                          \
      int foa = frag_off_a, fob = frag_off_b;                                                                  \
      asm volatile("" : "+v"(foa), "+v"(fob));                                                                 \
      nb0 = nbase + (uint32_t)(kOffB + wn * 64 * 128) + (uint32_t)fob;                                         \
      na0 = nbase + (uint32_t)(wm * (MS * 32) * 128) + (uint32_t)foa;                                          \
      if constexpr (WIDE) { /* held: the wave's 64 activation rows; streamed: its weight slots */              \
        nb0 = nbase + (uint32_t)(wm * 64 * 128) + (uint32_t)foa;                                               \
        na0 = nbase + (uint32_t)(kOffB + wn * 128 * 128) + (uint32_t)fob;                                      \
      }                                                                                                        \
      asm volatile("" : "+v"(nb0), "+v"(na0));                                                                 \
    }                                                                                                          \
    MP_STEP(MS - 1, STORE, true, false)                                                                        \
    if constexpr (!WIDE) { raw_c = raw_n;  sraw_c = sraw_n; }                                                  \
    if constexpr (NST == 3) { raw_n = raw_nn;  sraw_n = sraw_nn; }                                             \
    ++gblk;                                                                                                    \
    stg = s1;                                                                                                  \
  }

  int unit = 0;
  MpTile cur_t = describe(0);
  MpTile prv = describe(n_units);  // the null tile: nothing to store yet
  // ---- prologue: block 0 of the first unit lands (int4: is expanded), its fragments are read in the last step's order,
  // part 0 of block 1 goes out (int4: the codes of block 1 are requested)
#pragma unroll
  for (int part = 0; part < (WIDE ? 6 : 4); ++part) {
    dma_piece(cur_t, 0, 0, part, 0);
    dma_piece(cur_t, 0, 0, part, 1);
  }
  if constexpr (W4) {
    load_raw(cur_t, 0, raw_c, sraw_c);
    load_raw2(cur_t, 0, raw_c2, sraw_c2);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw_c), "+v"(sraw_c), "+v"(raw_c2), "+v"(sraw_c2));
#pragma unroll
    for (int q = 0; q < 4; ++q) expand(raw_c, sraw_c, q, lds_base + (uint32_t)kOffB + pwr);
    if constexpr (WIDE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) expand(raw_c2, sraw_c2, q, lds_base + (uint32_t)(kOffB + 256 * 128) + pwr);
    }
    load_raw(cur_t, 1, raw_c, sraw_c);
    load_raw2(cur_t, 1, raw_c2, sraw_c2);
    if constexpr (NST == 3) load_raw(cur_t, 2, raw_n, sraw_n);
  }
  if constexpr (NST == 3) {  // (all of block 1 and the a pieces of block 2 go out before anything waits: nkb >= 3)
    dma_piece(cur_t, 1, 1, 0, 0);  dma_piece(cur_t, 1, 1, 0, 1);
    dma_piece(cur_t, 1, 1, 2, 0);  dma_piece(cur_t, 1, 1, 2, 1);
    dma_piece(cur_t, 1, 1, 3, 0);  dma_piece(cur_t, 1, 1, 3, 1);
    dma_piece(cur_t, 2, 2, 0, 0);  dma_piece(cur_t, 2, 2, 0, 1);
  }
  // (three stages: block 0 has landed when only the pieces of blocks 1 and 2 - and the code loads - are outstanding; the wait
  // is simply for everything: once per workgroup)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    // (b0: the held fragments, p0: the first streamed one)
    const uint32_t b0 = WIDE ? lds_base + (uint32_t)(wm * 64 * 128) + (uint32_t)frag_off_a
                             : lds_base + (uint32_t)(kOffB + wn * 64 * 128) + (uint32_t)frag_off_b;
    const uint32_t p0 = WIDE ? lds_base + (uint32_t)(kOffB + wn * 128 * 128) + (uint32_t)frag_off_b
                             : lds_base + (uint32_t)(wm * (MS * 32) * 128) + (uint32_t)frag_off_a;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t x = (uint32_t)(s << 5);
      MP_RD16(nq[0][s], b0 ^ x, 0);
      MP_RD16(nq[1][s], b0 ^ x, 4096);
      MP_RD16(mq[s], p0 ^ x, 0);
    }
  }
  if constexpr (NST == 2) {
    dma_piece(cur_t, 1, 1, 0, 0);
    dma_piece(cur_t, 1, 1, 0, 1);
    dma_piece(cur_t, 1, 1, 1, 0);
    dma_piece(cur_t, 1, 1, 1, 1);
  }

  for (; unit < n_units; ++unit) {
    const MpTile nxt = describe(unit + 1);
    {
      const int kb = 0;
      MP_BLOCK(true)
    }
    for (int kb = 1; kb < nkb; ++kb) MP_BLOCK(false)
    prv = cur_t;
    cur_t = nxt;
  }
#undef MP_BLOCK
#undef MP_STEP
  // drain the reads and DMA of the last step; the MFMAs of the last step retire; store the last unit
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[MS - 1][0]), "+v"(acc[MS - 1][1]));
#pragma unroll
  for (int mf = 0; mf < MS; ++mf) {
    if constexpr (WIDE) store_frag_w(prv, acc[mf], acc[mf < 2 ? mf + 2 : mf], mf);
    else store_frag(prv, acc[mf], mf);
  }
  if (stamps_ != nullptr && lane == 0) {
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (wave == 0) {
      uint32_t* o = stamps_ + blockIdx.x * 4;
      o[0] = (uint32_t)(c1 - st_c0);  o[1] = (uint32_t)(r1 - st_r0);  o[2] = (uint32_t)(n_units * nkb);  o[3] = MS;
    }
    uint32_t* w = stamps_ + 1024 + (blockIdx.x * 8 + wave) * 2;  // per wave: cycles waiting for its own data / at the barrier
    w[0] = st_own;  w[1] = st_bar;
  }
#undef MP_MFMA
#undef MP_MFMA_BIAS
#undef MP_RD16
#undef MP_WR16
}

#ifdef SGLK_PROBES
static uint32_t* g_mp_stamps = nullptr;  // sglk_debug_set_moe_clock_stamps
static int g_mp_prio47 = 0;  // (sglk_debug_set_moe_prio: 1..3 swaps which half waits, the block time stays - DESIGN 4.10)
#else
constexpr uint32_t* g_mp_stamps = nullptr;
constexpr int g_mp_prio47 = 0;
#endif
#ifdef SGLK_PROBES
static int g_mp_own_tails = 0;
static int g_mp_wide = 1;
static int g_mp_splitk = 1;  // (sglk_debug_set_moe_splitk: 0 = the K split of the down projection off, for A / B timing)
#else
constexpr int g_mp_own_tails = 0;
constexpr int g_mp_wide = 1;
#endif
// (round 5, late - a token sweep of fused_experts across these boundaries: 383 tokens on the streaming kernels 645 us against 552 at
//  384 on the tiles, whose time is flat in the row count - the boundary moved from 96 to 88; and 129 - 191 rows per expert on 128-row
//  blocks left EVERY expert a remainder for the streaming launches - 640 / 704 / 767 tokens 0.98 / 1.07 / 1.06 ms against 0.73 / 0.745 /
//  0.76 on 256-row blocks, one partly filled block per expert: 256-row blocks now start at an average of 152 rows, not 192)
constexpr int kMinAvgRows128 = 88;  // ... and with 128-row blocks
#ifdef SGLK_PROBES
static int g_mp_min_avg_rows = 152;
#define kMinAvgRows g_mp_min_avg_rows
#else
constexpr int kMinAvgRows = 152;  // average rows per expert from which the tile pipeline takes over with 256-row blocks
#endif


template <typename T, int W4, bool BIAS = false>
static int launch_persist(hipStream_t st, const MpParams& p) {
  static unsigned long long attr_done4 = 0, attr_done2 = 0;
  constexpr int kLds2 = 3 * (kTile / 2 + kTile);  // 128-row blocks: three stages of 48 KiB
  if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&moe_persist_kernel<T, W4, 4, BIAS>), 2 * kStage, &attr_done4, "moe_persist"))
    return rc;
  if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&moe_persist_kernel<T, W4, 2, BIAS>), kLds2, &attr_done2, "moe_persist"))
    return rc;
  if (p.blocks128) {  // 96 .. 191 rows per expert on average: 128-row blocks - as 128 x 512 tiles (WIDE); the diagnostic build
                      // can fall back to the 128 x 256 tiles of MS = 2 for A / B timing
    // (4-bit weights with a bias: the wide form needs 2 .. 6 registers more than a lane has - those stay on the MS = 2 form)
    if constexpr (!(BIAS && W4 != 0)) if (g_mp_wide && p.blocks128 == 2) {
      static unsigned long long attr_donew = 0;
      constexpr int kLdsW = 2 * (128 * kBKB + 512 * kBKB);  // all 160 KiB
      if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&moe_persist_kernel<T, W4, 4, BIAS, true>), kLdsW, &attr_donew,
                                   "moe_persist"))
        return rc;
      moe_persist_kernel<T, W4, 4, BIAS, true><<<(unsigned)num_cus(), 512, kLdsW, st>>>(p);
      return 0;
    }
    moe_persist_kernel<T, W4, 2, BIAS><<<(unsigned)num_cus(), 512, kLds2, st>>>(p);
    return 0;
  }
  moe_persist_kernel<T, W4, 4, BIAS><<<(unsigned)num_cus(), 512, 2 * kStage, st>>>(p);
  // Remainders of at most 128 rows: the callers run their streaming kernels over them (moe_tiles.h, tail mode) - a few
  // dozen rows per expert are a weight stream, 70 us for the Mixtral down projection against 275 us as 128-row tiles here
  // (K = 14336: 96 long tiles on 256 CUs). MOE_PERSIST_TAILS=1 in the diagnostic build runs them here instead.
  if (g_mp_own_tails) moe_persist_kernel<T, W4, 2, BIAS><<<(unsigned)num_cus(), 512, kLds2, st>>>(p);
  return 0;
}

}  // namespace

// Called by sglk_moe_grouped_mm / sglk_moe_grouped_mm_w4a16_act. Returns 0 when the shape does not qualify (the caller goes
// on with its streaming kernels), 1 after launching 256-row blocks (the caller runs the tails in kMoeTailFlag mode), 2 after
// launching 128-row blocks (kMoeTailFlag128), 3 after launching 128-row blocks that cover every row (no tails), a negative error
// code on failure.
int moe_persist_try(hipStream_t st, void* out, const void* act, const void* w, const void* scales, const void* zeros,
                    int group_shift, const float* bias,
                    const int32_t* rows, int64_t total_m, int E, int N, int K, int64_t ldb, int64_t stride_e, int dtype, int w4,
                    int fuse, float act_limit, float act_alpha) {
  const bool gated = fuse == 1 || fuse == 2 || fuse == 4 || fuse == 5;
  const int Nout = gated ? N / 2 : N;
  if (total_m < (int64_t)kMinAvgRows128 * E || num_cus() % 8 != 0 || (uintptr_t)bias % 4 != 0) return 0;
  const bool blocks128 = total_m < (int64_t)kMinAvgRows * E;
  if (blocks128 && K < 192) return 0;  // (the three-stage ring of the 128-row blocks looks three K blocks ahead)
  if (w4 == 2 && dtype != SGLK_BF16) return 0;  // (the fp4 conversion instruction is used in its bf16 form)
  if (K % 64 != 0 || K < 128 || N % 8 != 0 || (gated && (N % 64 != 0)) || (uintptr_t)out % 16 != 0 ||
      (uintptr_t)act % 16 != 0 || (uintptr_t)w % 16 != 0 || Nout % 8 != 0)
    return 0;
  const int64_t b_row = w4 ? K / 2 : ldb * 2;
  if ((!w4 && (ldb % 8 != 0 || stride_e % 8 != 0)) || (int64_t)N * b_row >= (1ll << 32) || 264ll * K * 2 >= (1ll << 32) ||
      256ll * Nout * 2 + 512 >= (1ll << 31) ||
      (w4 == 1 && (group_shift < 5 || group_shift > 8 || (uintptr_t)scales % 2 != 0 || (uintptr_t)zeros % 2 != 0 ||
                   (int64_t)N * (K >> group_shift) * 2 >= (1ll << 31))) ||
      (w4 == 2 && (int64_t)N * (K / 32) >= (1ll << 31)))
    return 0;
  MpParams p;
  // 128-row blocks: as 128 x 512 tiles when that still gives every CU most of a tile (Mixtral gate / up at 512 tokens: 448 tiles;
  // its down projection - 8 column blocks x 8 row blocks = 64 tiles of 224 K blocks - stays on the 128 x 256 tiles: 128 of them)
  const int64_t wide_tiles = (total_m / 128) * ((Nout + (gated ? 255 : 511)) / (gated ? 256 : 512));
  p.blocks128 = !blocks128 ? 0 : wide_tiles >= 192 ? 2 : 1;
  // Round 5: when even the worst case of 128 x 256 tiles - every expert with a remainder block - fits ONE round of the CUs, the
  // remainders of 1 .. 64 rows run here as partly empty blocks on CUs that would idle, instead of a tail launch that streams those
  // experts' weights a second time (Mixtral down projection at 512 tokens: 128 - 256 tiles of 224 K blocks on 256 CUs; the tail
  // launch cost ~60 us of the layer's 750).
  const int64_t nb_cols = gated ? (Nout + 127) / 128 : (N + 255) / 256;
  p.own_rem = (p.blocks128 == 1 && (total_m / 128 + E) * nb_cols <= (int64_t)num_cus()) ? 1 : 0;
  p.stamps = g_mp_stamps;
  p.prio47 = g_mp_prio47;
  p.out = out;  p.act = act;  p.w = w;  p.scales = scales;  p.zeros = zeros;  p.bias = bias;  p.gshift = group_shift;  p.rows = rows;
  p.total_m = total_m;
  p.E = E;  p.N = N;  p.K = K;  p.fuse = fuse == 5 ? 4 : fuse;  p.pairs = fuse == 5 ? 1 : 0;  p.act_limit = act_limit;
  p.act_kneg = -1.4426950408889634f * (fuse == 5 ? act_alpha : 1.0f);  p.act_yadd = fuse == 5 ? 1.0f : 0.0f;
  p.ldb = ldb;  p.stride_e = stride_e;
  int rc;
  const int fmt = w4 == 1 && zeros != nullptr ? 3 : w4;
  if (bias != nullptr) {
    if (dtype == SGLK_BF16)
      rc = fmt == 3 ? launch_persist<bf16, 3, true>(st, p) : fmt == 2 ? launch_persist<bf16, 2, true>(st, p)
           : fmt == 1 ? launch_persist<bf16, 1, true>(st, p) : launch_persist<bf16, 0, true>(st, p);
    else
      rc = fmt == 3 ? launch_persist<f16, 3, true>(st, p) : fmt == 1 ? launch_persist<f16, 1, true>(st, p)
                                                           : launch_persist<f16, 0, true>(st, p);
  } else if (dtype == SGLK_BF16)
    rc = fmt == 3 ? launch_persist<bf16, 3>(st, p) : fmt == 2 ? launch_persist<bf16, 2>(st, p)
         : fmt == 1 ? launch_persist<bf16, 1>(st, p) : launch_persist<bf16, 0>(st, p);
  else
    rc = fmt == 3 ? launch_persist<f16, 3>(st, p) : fmt == 1 ? launch_persist<f16, 1>(st, p) : launch_persist<f16, 0>(st, p);
  return rc ? rc : (blocks128 ? (p.own_rem ? 3 : 2) : 1);
}

// ---- K split of the 128- / 256-row blocks (4-bit weights, no bias, no activation: the down projection of fused_experts) ----
// Applies when the tiles of the FULL row blocks (128 x 256 below an average of 192 rows per expert, 256 x 256 from there) are
// at most 9/16 of the CUs, so that two units per tile fit one round (an expert's remainder of more than half a block is a block
// too: up to E more row blocks, i.e. a second round of half-length units at worst - never more K blocks per CU than without the
// split), K halves are whole scale groups and at least four K blocks long. Remainders of up to half a block stay with the
// caller's streaming kernels (kMoeTailFlag128 / kMoeTailFlag) and go to `out`. Returns the row block (128 / 256) or 0.
int moe_persist_splitk_applies(int64_t total_m, int E, int N, int K, int group_shift, int w4, int dtype) {
#ifdef SGLK_PROBES
  if (g_mp_splitk == 0) return 0;
#endif
  if (num_cus() % 8 != 0 || E <= 0) return 0;
  if (total_m < (int64_t)kMinAvgRows128 * E) return 0;  // (below: the streaming kernels)
  const int block = total_m < (int64_t)kMinAvgRows * E ? 128 : 256;
  if (w4 != 1 && w4 != 2) return 0;
  if (w4 == 2 && dtype != SGLK_BF16) return 0;
  const int gsh = w4 == 2 ? 5 : group_shift;
  if (K % 128 != 0 || (K / 2) % (1 << gsh) != 0 || K / 128 < 4) return 0;
  if (N % 8 != 0) return 0;
  const int64_t nb_cols = (N + 255) / 256;
  return (total_m / block) * nb_cols * 16 <= (int64_t)num_cus() * 9 ? block : 0;
}

// Launches the split form over the full row blocks; returns 0 when it does not apply, 2 after launching 128-row blocks (the caller
// runs the remainders of 1 .. 64 rows in kMoeTailFlag128 mode into its 16-bit `out`), 1 after launching 256-row blocks (remainders
// of 1 .. 128 rows, kMoeTailFlag), a negative error code on failure.
int moe_persist_splitk_try(hipStream_t st, float* ws, const void* act, const void* w, const void* scales, const void* zeros,
                           int group_shift, const int32_t* rows, int64_t total_m, int E, int N, int K, int dtype, int w4) {
  const int block = moe_persist_splitk_applies(total_m, E, N, K, group_shift, w4, dtype);
  if (block == 0) return 0;
  if ((uintptr_t)ws % 16 != 0 || (uintptr_t)act % 16 != 0 || (uintptr_t)w % 16 != 0) return 0;
  const int64_t b_row = K / 2;
  if ((int64_t)N * b_row >= (1ll << 32) || 264ll * K * 2 >= (1ll << 32) || 256ll * N * 4 + 1024 >= (1ll << 31) ||
      (w4 == 1 && (group_shift < 5 || group_shift > 8 || (uintptr_t)scales % 2 != 0 || (uintptr_t)zeros % 2 != 0 ||
                   (int64_t)N * (K >> group_shift) * 2 >= (1ll << 31))) ||
      (w4 == 2 && (int64_t)N * (K / 32) >= (1ll << 31)))
    return 0;
  MpParams p;
  p.blocks128 = block == 128 ? 1 : 0;
  p.own_rem = 0;
  p.stamps = g_mp_stamps;
  p.prio47 = g_mp_prio47;
  p.out = ws;  p.act = act;  p.w = w;  p.scales = scales;  p.zeros = zeros;  p.bias = nullptr;  p.gshift = group_shift;  p.rows = rows;
  p.total_m = total_m;
  p.E = E;  p.N = N;  p.K = K;  p.fuse = 0;  p.pairs = 0;  p.act_limit = 0.f;  p.act_kneg = 0.f;  p.act_yadd = 0.f;  p.ldb = 0;  p.stride_e = 0;
  constexpr int kLds2 = 3 * (kTile / 2 + kTile);
  const int fmt = w4 == 1 && zeros != nullptr ? 3 : w4;
#define MP_GO_SPLIT(TT, FF)                                                                                         \
  {                                                                                                                 \
    static unsigned long long attr_done = 0, attr_done4 = 0;                                                        \
    if (block == 128) {                                                                                             \
      if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&moe_persist_kernel<TT, FF, 2, false, false, 2>), kLds2, \
                                   &attr_done, "moe_persist"))                                                      \
        return rc;                                                                                                  \
      moe_persist_kernel<TT, FF, 2, false, false, 2><<<(unsigned)num_cus(), 512, kLds2, st>>>(p);                   \
    } else {                                                                                                        \
      if (int rc = set_max_dyn_lds(reinterpret_cast<const void*>(&moe_persist_kernel<TT, FF, 4, false, false, 2>), 2 * kStage, \
                                   &attr_done4, "moe_persist"))                                                     \
        return rc;                                                                                                  \
      moe_persist_kernel<TT, FF, 4, false, false, 2><<<(unsigned)num_cus(), 512, 2 * kStage, st>>>(p);              \
    }                                                                                                               \
  }
  if (dtype == SGLK_BF16) {
    if (fmt == 3) MP_GO_SPLIT(bf16, 3) else if (fmt == 2) MP_GO_SPLIT(bf16, 2) else MP_GO_SPLIT(bf16, 1)
  } else {
    if (fmt == 3) MP_GO_SPLIT(f16, 3) else MP_GO_SPLIT(f16, 1)
  }
#undef MP_GO_SPLIT
  return block == 128 ? 2 : 1;
}

}  // namespace sglk

#ifdef SGLK_PROBES
extern "C" SGLK_API void sglk_debug_set_moe_splitk(int on) { sglk::g_mp_splitk = on; }
// Diagnostic build only. Clock stamps of the tile pipeline: 256 x 4 uint32, one record per workgroup of the last launch
// {shader cycles, 100 MHz ticks, K blocks, m-steps}, followed by 256 x 8 x 2 uint32: per wave the shader cycles spent in
// front of the K blocks' barriers waiting for its own LDS-DMA / LDS data, and at the barriers themselves (5120 uint32).
extern "C" SGLK_API void sglk_debug_set_moe_clock_stamps(uint32_t* device_buf) { sglk::g_mp_stamps = device_buf; }
extern "C" SGLK_API void sglk_debug_set_moe_prio(int prio) { sglk::g_mp_prio47 = prio; }
extern "C" SGLK_API void sglk_debug_set_moe_persist_min_rows(int rows) { sglk::g_mp_min_avg_rows = rows; }
extern "C" SGLK_API void sglk_debug_set_moe_persist_own_tails(int on) { sglk::g_mp_own_tails = on; }
extern "C" SGLK_API void sglk_debug_set_moe_persist_wide(int on) { sglk::g_mp_wide = on; }
#endif
