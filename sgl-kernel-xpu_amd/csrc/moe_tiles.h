// Workgroup -> (expert, block of its rows, column block) for the grouped GEMMs over ragged expert row lists
// (moe_w4a16.hip, moe_bf16.hip). The row counts live on the device (no host sync), so the launch is one-dimensional
// and sized for the worst case: sum_e ceil(rows_e / BM) <= total_rows / BM + E row blocks, times NB column blocks,
// rounded up to a multiple of 8.
//
// Each workgroup sums the actual number MB of row blocks from rows_per_expert (lanes = experts, wave prefix sums:
// ~0.2 us against 1-2 us for a scalar walk over dependent loads) and maps its launch index L to (row block L % MB,
// column block L / MB): the row blocks of one column block - they share its weights - are neighbours in launch order
// and the unused indices all lie at the END of the launch. With a 2-D grid of max-row-blocks x column-blocks the empty
// workgroups sat in between: they retire at once, the dispatcher hands the next real workgroup to whichever CU is free,
// and a third of the CUs ended up with two weight streams while others had none (350 us instead of 185 us for 16 rows
// per expert at N = 28672, K = 4096). Workgroups go to the 8 XCDs round-robin in launch order, so the index is
// re-ordered (within the real tiles) to keep neighbours on one XCD (one L2 for the shared weights).
#pragma once
#include "common.h"

namespace sglk {

struct MoeTile {
  int expert;     // -1: nothing to do
  int m0;         // first global row of the block
  int m_valid;    // rows of the block that exist (>= 1)
  int col_block;
};

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// every wave of the workgroup computes the same (wave-uniform) answer
// cols_fast: consecutive tiles (= the workgroups resident on one XCD at a time) walk the column blocks of ONE row block and
// share its activations in that XCD's L2; otherwise they walk the row blocks of one column block. The larger operand of a
// tile should be the shared one: 64 rows x K 16-bit activations against 128 columns x K / 2 bytes of 4-bit weights.
// Tail mode (bit 30 or 29 of E: kMoeTailFlag / kMoeTailFlag128): the launch covers only what the tile pipeline of
// moe_persist.hip leaves over - of every expert the rows behind its last full block of B = 256 (128) rows when they are at
// most B / 2 (more are a block of their own there).
constexpr int kMoeTailFlag = 1 << 30, kMoeTailFlag128 = 1 << 29;
__host__ __device__ __forceinline__ int moe_tail_rows(int r, int B) {
  const int t = r & (B - 1);
  return t <= B / 2 ? t : 0;
}
__device__ __forceinline__ MoeTile find_moe_tile(const int32_t* __restrict__ rows_per_expert, int E, int BM, int NB,
                                                 bool cols_fast = false) {
  const int lane = threadIdx.x & 63;
  const bool tail_mode = (E & (kMoeTailFlag | kMoeTailFlag128)) != 0;
  const int tail_b = (E & kMoeTailFlag) ? 256 : 128;
  E &= ~(kMoeTailFlag | kMoeTailFlag128);
  int MB = 0;
  for (int c0 = 0; c0 < E; c0 += 64) {
    const int r = c0 + lane < E ? rows_per_expert[c0 + lane] : 0;
    const int rr = tail_mode ? moe_tail_rows(r, tail_b) : r;
    MB += __shfl(wave_inclusive_scan((rr + BM - 1) / BM, lane), 63, 64);
  }
  MB = __builtin_amdgcn_readfirstlane(MB);
  MoeTile t = {-1, 0, 0, 0};
  // launch index -> tile index: XCD x (launch indices = x mod 8) takes a contiguous eighth of the REAL tiles (the unused
  // tail of the launch must stay out of the re-ordering, or the XCDs that get it sit idle)
  const int real = MB * NB, real8 = real & ~7;
  int L = blockIdx.x;
  if (L >= real) return t;
  if (L < real8) L = (L & 7) * (real8 >> 3) + (L >> 3);
  const int mblk = cols_fast ? L / NB : L % MB;
  int e = 0, row0 = 0, rows_e = 0, blk = 0, base_b = 0, base_r = 0;
  for (int c0 = 0; c0 < E; c0 += 64) {
    const int r = c0 + lane < E ? rows_per_expert[c0 + lane] : 0;
    const int rr = tail_mode ? moe_tail_rows(r, tail_b) : r;
    const int nb = (rr + BM - 1) / BM;
    const int ib = wave_inclusive_scan(nb, lane), ir = wave_inclusive_scan(r, lane);
    const bool hit = mblk >= base_b + ib - nb && mblk < base_b + ib;
    const unsigned long long m = __ballot(hit);
    if (m != 0) {
      const int src = __builtin_ctzll(m);
      e = c0 + src;
      blk = mblk - (base_b + __shfl(ib - nb, src, 64));
      row0 = base_r + __shfl(ir - r, src, 64) + __shfl(r - rr, src, 64);  // (tail mode: the expert's first rows are not ours)
      rows_e = __shfl(rr, src, 64);
      break;
    }
    base_b += __shfl(ib, 63, 64);
    base_r += __shfl(ir, 63, 64);
  }
  t.expert = __builtin_amdgcn_readfirstlane(e);
  blk = __builtin_amdgcn_readfirstlane(blk);
  t.m0 = __builtin_amdgcn_readfirstlane(row0) + blk * BM;
  t.m_valid = __builtin_amdgcn_readfirstlane(rows_e) - blk * BM;
  t.col_block = cols_fast ? L % NB : L / MB;
  return t;
}

// launch size for find_moe_tile
inline int64_t moe_tile_launch_size(int64_t total_rows, int64_t E, int64_t BM, int64_t NB) {
  return ((total_rows / BM + E) * NB + 7) / 8 * 8;
}

}  // namespace sglk
