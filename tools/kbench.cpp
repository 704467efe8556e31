// Standalone (torch-free) timing harness for the C-ABI kernels, linked against the DIAGNOSTIC library
// (build/libsglk_probes.so, -DSGLK_PROBES: main-loop variants and garbage-result timing probes behind sglk_debug_*):
//   python sgl-kernel-xpu_amd/build.py --probes   ->  sgl-kernel-xpu_amd/build/kbench
// usage: kbench gemm M N K [variants...] | kbench stamps M N K | kbench scaledmm M N K | kbench mla B S H [splits...]
//        kbench peak THREADS BLOCKS ITERS | kbench oob | kbench w4a16 N K ROWS_PER_EXPERT [probe:mt ...]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "sglk.h"

extern "C" void sglk_debug_set_gemm_variant(int);
extern "C" void sglk_debug_set_mla_waves_per_group(int);
extern "C" void sglk_debug_set_mla_probe(int);
extern "C" void sglk_debug_set_mla_variant(int);
extern "C" int sglk_debug_get_mla_stamps(unsigned long long*, int);
extern "C" void sglk_debug_set_gemm_stamps(uint32_t*);
extern "C" void sglk_debug_set_gemm_stagger(int);
extern "C" void sglk_debug_set_skinny_la_rows(int);
extern "C" void sglk_debug_set_w4a16_probe(int probe, int force_mt);

#define HIP_CHECK(x)                                                                 \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

template <typename F>
static double time_ms(F&& f, int warm, int iters, std::vector<float>* all = nullptr) {
  for (int i = 0; i < warm; ++i) f();
  HIP_CHECK(hipDeviceSynchronize());
  std::vector<hipEvent_t> ev(iters + 1);
  for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipEventRecord(ev[0], 0));
  for (int i = 0; i < iters; ++i) {
    f();
    HIP_CHECK(hipEventRecord(ev[i + 1], 0));
  }
  HIP_CHECK(hipDeviceSynchronize());
  std::vector<float> ms(iters);
  for (int i = 0; i < iters; ++i) HIP_CHECK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  std::sort(ms.begin(), ms.end());
  if (all) *all = ms;
  return ms[iters / 2];
}

static void* dev_random_bytes(size_t n, unsigned seed, bool fp8_safe) {
  std::vector<uint8_t> h(n);
  std::mt19937 rng(seed);
  for (size_t i = 0; i < n; i += 4) {
    uint32_t r = rng();
    for (int j = 0; j < 4 && i + j < n; ++j) {
      uint8_t b = (uint8_t)(r >> (8 * j));
      if (fp8_safe && (b & 0x7f) == 0x7f) b ^= 1;  // no e4m3 NaN
      h[i + j] = b;
    }
  }
  void* d;
  HIP_CHECK(hipMalloc(&d, n));
  HIP_CHECK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
  return d;
}
static float* dev_random_floats(size_t n, unsigned seed, float lo, float hi) {
  std::vector<float> h(n);
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> d01(lo, hi);
  for (auto& x : h) x = d01(rng);
  float* d;
  HIP_CHECK(hipMalloc(&d, n * 4));
  HIP_CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}


// ---- MFMA ceiling probe: no memory traffic in the loop, operands from registers --------------------------------
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void mfma_peak_kernel(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  v8i_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) a[i][j] = src[(threadIdx.x * 8 + j + i * 4096) & 16383];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) b[i][j] = src[(threadIdx.x * 8 + j + i * 4096 + 777) & 16383];
  if constexpr (MODE == 3) {  // bf16 16x16x32, 16 independent accumulators
    typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    v4f_t acc[4][4] = {};
    v4i_t ah[4], bh[4];
    for (int i = 0; i < 4; ++i) { ah[i] = (v4i_t){a[i][0], a[i][1], a[i][2], a[i][3]}; bh[i] = (v4i_t){b[i][0], b[i][1], b[i][2], b[i][3]}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, bh[n]), __builtin_bit_cast(v8bf_t, ah[m]), acc[m][n], 0, 0, 0);
      asm volatile("" : "+v"(ah[0][0]), "+v"(ah[1][0]), "+v"(ah[2][0]), "+v"(ah[3][0]));
    }
    float s = 0;
    for (int m = 0; m < 4; ++m)
      for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else if constexpr (MODE == 2) {
    v16f_t acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b[n], a[u + 2 * (it & 1)], acc[n], 0, 0, 0, 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < 4; ++n)
      for (int r = 0; r < 16; ++r) s += acc[n][r];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    v4f_t acc[4][4] = {};
    const v4f_t zero = {0, 0, 0, 0};
    float sc = __int_as_float(0x3f800000 | (src[threadIdx.x] & 0xffff));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if constexpr (MODE == 0) {
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[n], a[m], acc[m][n], 0, 0, 0, 0, 0, 0);
        } else {
          v4f_t cur[4];
#pragma unroll
          for (int n = 0; n < 4; ++n)
            cur[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[n], a[m], zero, 0, 0, 0, 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] = __builtin_fmaf(cur[n][r], sc, acc[m][n][r]);
        }
      }
      asm volatile("" : "+v"(sc), "+v"(a[0][0]), "+v"(a[1][0]), "+v"(a[2][0]), "+v"(a[3][0]));
    }
    float s = 0;
    for (int m = 0; m < 4; ++m)
      for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

// ---- issue-cost probe: the GEMM main loop's instruction mix without memory (operands in registers): zero-C MFMA of one
// 128-deep block per 16x16 (or 32x32) tile and the block-scale promotion of the PREVIOUS tile's partial between MFMAs
//   0: 16x16x128 + 4 v_fma_f32    1: 16x16x128 + 2 v_pk_fma_f32    2: 2 x 32x32x64 + 16 v_fma_f32
//   3: 2 x 32x32x64 + 8 v_pk_fma_f32    4: 16x16x128 alone    5: 2 x 32x32x64 alone
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void issue_probe_kernel(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  v8i_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = src[(threadIdx.x * 8 + j + i * 4096) & 16383];
      b[i][j] = src[(threadIdx.x * 8 + j + i * 4096 + 777) & 16383];
    }
  int one = 127;
  float sc = __int_as_float(0x3f800000 | (src[threadIdx.x] & 0xffff));
  v2f_t sc2 = {sc, sc};
  asm volatile("" : "+v"(one), "+v"(sc), "+v"(sc2));
  float s = 0;
  if constexpr (MODE == 0 || MODE == 1 || MODE == 4) {
    float acc[16][4] = {};
    v2f_t acc2[16][2] = {};
    v4f_t cur[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"
                     : "=&v"(cur[t & 1]) : "v"(b[t & 3]), "v"(a[t >> 2]), "v"(one));
        constexpr int dummy = 0; (void)dummy;
        const int p = (t + 15) & 15;
        if constexpr (MODE == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[p][r]) : "v"(cur[(t & 1) ^ 1][r]), "v"(sc));
        } else if constexpr (MODE == 1) {
          const v2f_t lo = {cur[(t & 1) ^ 1][0], cur[(t & 1) ^ 1][1]}, hi = {cur[(t & 1) ^ 1][2], cur[(t & 1) ^ 1][3]};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[p][0]) : "v"(lo), "v"(sc2));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[p][1]) : "v"(hi), "v"(sc2));
        }
      }
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(cur[0]), "+v"(cur[1]));
    for (int t = 0; t < 16; ++t)
      for (int r = 0; r < 4; ++r) s += acc[t][r] + acc2[t][r >> 1][r & 1];
    s += cur[0][0] + cur[1][0];
  } else {
    float acc[4][16] = {};
    v2f_t acc2[4][8] = {};
    v16f_t cur[2];
    for (int i = 0; i < 16; ++i) { cur[0][i] = 0; cur[1][i] = 0; }
    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"
                     : "=&v"(cur[t & 1]) : "v"(b[t & 3]), "v"(a[t >> 1]), "v"(one));
        const int p = (t + 3) & 3;
        if constexpr (MODE == 2) {
#pragma unroll
          for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[p][r]) : "v"(cur[(t & 1) ^ 1][r]), "v"(sc));
        } else if constexpr (MODE == 3) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const v2f_t x = {cur[(t & 1) ^ 1][2 * r], cur[(t & 1) ^ 1][2 * r + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[p][r]) : "v"(x), "v"(sc2));
          }
        }
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                     : "+v"(cur[t & 1]) : "v"(a[t & 3]), "v"(b[t >> 1]), "v"(one));
        if constexpr (MODE == 2) {
#pragma unroll
          for (int r = 8; r < 16; ++r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[p][r]) : "v"(cur[(t & 1) ^ 1][r]), "v"(sc));
        } else if constexpr (MODE == 3) {
#pragma unroll
          for (int r = 4; r < 8; ++r) {
            const v2f_t x = {cur[(t & 1) ^ 1][2 * r], cur[(t & 1) ^ 1][2 * r + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[p][r]) : "v"(x), "v"(sc2));
          }
        }
      }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(cur[0]), "+v"(cur[1]));
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 16; ++r) s += acc[t][r] + acc2[t][r >> 1][r & 1];
    s += cur[0][0] + cur[1][0];
  }
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- does the raw-buffer range check of buffer_load ... lds include the scalar offset? (expects zeros past the end)
// Weight-streaming access patterns of a [rows x row_bytes] matrix (the int4 expert weights): every workgroup streams 128
// rows end to end, 4 waves x 32 rows, 16 bytes per lane and load, DEPTH loads in flight per lane.
//   pattern 0: one load = 16 rows x 64 contiguous bytes (lane = (row l15, chunk g)): the MFMA-fragment order
//   pattern 1: one load = 4 rows x 256 contiguous bytes (lane = (row lane / 16, chunk lane % 16))
//   pattern 2: one load = 1 row x 1024 contiguous bytes
// pattern 3: the W4A16 kernel's order - a round is ONE 64-byte piece of each of the wave's 32 rows (2 loads per lane),
// DEPTH rounds in flight, consecutive rounds advance along the rows (the next 64 bytes of a row come a round later)
template <int DEPTH, int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void stream_probe3_kernel(const uint8_t* __restrict__ w, uint32_t* __restrict__ out,
                                                                       int row_bytes) {
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
  v4i_ acc = {0, 0, 0, 0};
  const int rounds = row_bytes / 64;
  const uint8_t* p0 = w + (row0 + (lane & 15)) * row_bytes + (lane >> 4) * 16;
  const uint8_t* p1 = p0 + (int64_t)16 * row_bytes;
  v4i_ ring[DEPTH][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    ring[d][0] = *reinterpret_cast<const v4i_*>(p0 + d * 64);
    ring[d][1] = *reinterpret_cast<const v4i_*>(p1 + d * 64);
  }
  for (int r0 = 0; r0 < rounds; r0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc ^= ring[d][0] ^ ring[d][1];
      const int rn = r0 + d + DEPTH < rounds ? r0 + d + DEPTH : rounds - 1;
      ring[d][0] = *reinterpret_cast<const v4i_*>(p0 + rn * 64);
      ring[d][1] = *reinterpret_cast<const v4i_*>(p1 + rn * 64);
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[0] = 1;
}

template <int PATTERN, int DEPTH>
__global__ __launch_bounds__(256) void stream_probe_kernel(const uint8_t* __restrict__ w, uint32_t* __restrict__ out, int row_bytes) {
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
  v4i_ acc = {0, 0, 0, 0};
  // one "round" moves 32 rows x 256 bytes per wave = 8 loads per lane
  const int rounds = row_bytes / 256;
  for (int r0 = 0; r0 < rounds; r0 += DEPTH) {
    v4i_ v[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int r = r0 + d;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int64_t row, off;
        if (PATTERN == 0) {  // i = (n tile nt = i / 4, 64-byte block q = i % 4)
          row = row0 + (i >> 2) * 16 + (lane & 15);
          off = (int64_t)r * 256 + (i & 3) * 64 + (lane >> 4) * 16;
        } else if (PATTERN == 1) {  // i = 4-row group
          row = row0 + i * 4 + (lane >> 4);
          off = (int64_t)r * 256 + (lane & 15) * 16;
        } else {  // 8 loads = 8 rows x 1 KB; a round pair covers the 32 rows
          row = row0 + (r & 3) * 8 + i;
          off = (int64_t)(r >> 2) * 1024 + lane * 16;
        }
        v[d][i] = *reinterpret_cast<const v4i_*>(w + row * row_bytes + off);
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= v[d][i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[0] = 1;
}

__global__ void oob_probe_kernel(const uint8_t* __restrict__ src, int nrec, int soff, int use_voff, uint32_t* out) {
  __shared__ __attribute__((aligned(256))) uint32_t lds[256];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  if (use_voff)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, lane * 16 + soff, 0, 0, 0);
  else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, lane * 16, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  if (!strcmp(argv[1], "oob")) {
    uint8_t* src; uint32_t* out;
    HIP_CHECK(hipMalloc(&src, 8192)); HIP_CHECK(hipMemset(src, 0x11, 8192));
    HIP_CHECK(hipMalloc(&out, 1024));
    for (int use_voff = 0; use_voff < 2; ++use_voff) {
      oob_probe_kernel<<<1, 64>>>(src, 4096, 4096 - 512, use_voff, out);
      uint32_t h[256];
      HIP_CHECK(hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost));
      int in_ok = 0, out_zero = 0, out_data = 0, out_other = 0;
      for (int l = 0; l < 64; ++l)
        for (int d = 0; d < 4; ++d) {
          uint32_t v = h[l * 4 + d];
          if (l < 32) in_ok += (v == 0x11111111u);
          else { out_zero += (v == 0); out_data += (v == 0x11111111u); out_other += (v != 0 && v != 0x11111111u); }
        }
      printf("oob probe (%s carries the offset): in-range ok %d/128, past-the-end: zero %d data %d other %d of 128\n",
             use_voff ? "voffset" : "soffset", in_ok, out_zero, out_data, out_other);
    }
    return 0;
  }
  if (!strcmp(argv[1], "peak")) {  // kbench peak THREADS BLOCKS ITERS
    const int threads = atoi(argv[2]), blocks = atoi(argv[3]), iters = atoi(argv[4]);
    int* src = (int*)dev_random_bytes(16384 * 4, 9, true);
    float* dst;
    HIP_CHECK(hipMalloc(&dst, (size_t)blocks * threads * 4));
    for (int mode = 0; mode < 4; ++mode) {
      auto run = [&] {
        if (mode == 0) mfma_peak_kernel<0><<<blocks, threads>>>(src, dst, iters);
        if (mode == 1) mfma_peak_kernel<1><<<blocks, threads>>>(src, dst, iters);
        if (mode == 2) mfma_peak_kernel<2><<<blocks, threads>>>(src, dst, iters);
        if (mode == 3) mfma_peak_kernel<3><<<blocks, threads>>>(src, dst, iters);
      };
      std::vector<float> all;
      const double ms = time_ms(run, 3, 10, &all);
      const double flops = (double)blocks * (threads / 64) * iters * (mode == 2 ? 8 : 16) * (mode == 2 ? 32.0 * 32 * 64 * 2 : mode == 3 ? 16.0 * 16 * 32 * 2 : 16.0 * 16 * 128 * 2);
      printf("peak mode=%d threads=%d blocks=%d iters=%d: median %.4f ms min %.4f -> %.1f TFLOP/s\n", mode, threads, blocks,
             iters, ms, all[0], flops / ms / 1e9);
    }
    return 0;
  }
  if (!strcmp(argv[1], "issue")) {  // kbench issue THREADS BLOCKS ITERS
    const int threads = atoi(argv[2]), blocks = atoi(argv[3]), iters = atoi(argv[4]);
    int* src = (int*)dev_random_bytes(16384 * 4, 9, true);
    float* dst;
    HIP_CHECK(hipMalloc(&dst, (size_t)blocks * threads * 4));
    const char* names[6] = {"16x16x128 + 4 v_fma", "16x16x128 + 2 v_pk_fma", "2x 32x32x64 + 16 v_fma", "2x 32x32x64 + 8 v_pk_fma",
                            "16x16x128 alone", "2x 32x32x64 alone"};
    for (int mode = 0; mode < 6; ++mode) {
      auto run = [&] {
        if (mode == 0) issue_probe_kernel<0><<<blocks, threads>>>(src, dst, iters);
        if (mode == 1) issue_probe_kernel<1><<<blocks, threads>>>(src, dst, iters);
        if (mode == 2) issue_probe_kernel<2><<<blocks, threads>>>(src, dst, iters);
        if (mode == 3) issue_probe_kernel<3><<<blocks, threads>>>(src, dst, iters);
        if (mode == 4) issue_probe_kernel<4><<<blocks, threads>>>(src, dst, iters);
        if (mode == 5) issue_probe_kernel<5><<<blocks, threads>>>(src, dst, iters);
      };
      std::vector<float> all;
      const double ms = time_ms(run, 3, 10, &all);
      const double flops = (double)blocks * (threads / 64) * iters * 16 * 16.0 * 16 * 128 * 2;  // both shapes: 16 x 65536 per iteration
      printf("issue mode=%d (%s) threads=%d: median %.4f ms -> %.1f TFLOP/s\n", mode, names[mode], threads, ms, flops / ms / 1e9);
    }
    return 0;
  }
  if (!strcmp(argv[1], "gemm")) {
    const int64_t M = atoll(argv[2]), N = atoll(argv[3]), K = atoll(argv[4]);
    void* a = dev_random_bytes(M * K, 1, true);
    void* b = dev_random_bytes(N * K, 2, true);
    float* sa = dev_random_floats(M * (K / 128), 3, 1e-4f, 1e-3f);
    float* sb = dev_random_floats((K / 128) * ((N + 127) / 128), 4, 1e-4f, 1e-3f);
    void* out;
    HIP_CHECK(hipMalloc(&out, M * N * 2));
    for (int ai = 5; ai < argc || ai == 5; ++ai) {
      const int var = ai < argc ? atoi(argv[ai]) : 1;
      sglk_debug_set_gemm_variant(var);
      if (getenv("GEMM_STAGGER")) sglk_debug_set_gemm_stagger(atoi(getenv("GEMM_STAGGER")));
      if (getenv("GEMM_LA_ROWS")) sglk_debug_set_skinny_la_rows(atoi(getenv("GEMM_LA_ROWS")));
      std::vector<float> all;
      auto run = [&] {
        int rc = sglk_fp8_blockwise_scaled_mm(0, out, a, b, sa, sb, M, N, K, K, K, N, 1, M, 1, K / 128, SGLK_BF16);
        if (rc) { fprintf(stderr, "error: %s\n", sglk_last_error()); exit(1); }
      };
      const double ms = time_ms(run, 300, 100, &all);  // long warm-up: clocks ramp for tens of ms
      printf("gemm M=%lld N=%lld K=%lld variant=%d median %.4f ms  min %.4f  -> %.1f TFLOP/s (%.1f at min)\n", (long long)M,
             (long long)N, (long long)K, var, ms, all[0], 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / all[0] / 1e9);
      if (getenv("GEMM_CLOCK")) {  // the 32x32x64 kernel's own stamps: the clock each workgroup ran at, cycles per K block
        uint32_t* st;
        HIP_CHECK(hipMalloc(&st, 512 * 16));
        HIP_CHECK(hipMemset(st, 0, 512 * 16));
        sglk_debug_set_gemm_stamps(st);
        for (int r = 0; r < 20; ++r) run();
        HIP_CHECK(hipDeviceSynchronize());
        sglk_debug_set_gemm_stamps(nullptr);
        std::vector<uint32_t> h(512 * 4);
        HIP_CHECK(hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> mhz, cpb;
        for (int w = 0; w < 512; ++w)
          if (h[w * 4 + 1] && h[w * 4 + 3] == 4) {
            mhz.push_back(100.0 * h[w * 4] / h[w * 4 + 1]);
            cpb.push_back((double)h[w * 4] / h[w * 4 + 2]);
          }
        if (!mhz.empty()) {
          std::sort(mhz.begin(), mhz.end());
          std::sort(cpb.begin(), cpb.end());
          printf("  in-kernel (whole tiles, %zu workgroups): clock median %.0f MHz (min %.0f max %.0f); shader cycles per K block median %.0f"
                 " (MFMA time 2048)\n", mhz.size(), mhz[mhz.size() / 2], mhz.front(), mhz.back(), cpb[cpb.size() / 2]);
        }
        mhz.clear();  cpb.clear();
        for (int w = 0; w < 512; ++w)
          if (h[w * 4 + 1] && h[w * 4 + 3] == 2) {
            mhz.push_back(100.0 * h[w * 4] / h[w * 4 + 1]);
            cpb.push_back((double)h[w * 4] / h[w * 4 + 2]);
          }
        if (!mhz.empty()) {
          std::sort(mhz.begin(), mhz.end());
          std::sort(cpb.begin(), cpb.end());
          printf("  in-kernel (half tiles, %zu workgroups): clock median %.0f MHz; shader cycles per K block median %.0f (MFMA time 1024)\n",
                 mhz.size(), mhz[mhz.size() / 2], cpb[cpb.size() / 2]);
        }
        HIP_CHECK(hipFree(st));
      }
      if (ai >= argc) break;
    }
    return 0;
  }
  if (!strcmp(argv[1], "gemmab")) {  // kbench gemmab M N K variant:stagger ... - interleaved rounds of the listed forms on one device
    const int64_t M = atoll(argv[2]), N = atoll(argv[3]), K = atoll(argv[4]);
    void* a = dev_random_bytes(M * K, 1, true);
    void* b = dev_random_bytes(N * K, 2, true);
    float* sa = dev_random_floats(M * (K / 128), 3, 1e-4f, 1e-3f);
    float* sb = dev_random_floats((K / 128) * ((N + 127) / 128), 4, 1e-4f, 1e-3f);
    void* out;
    HIP_CHECK(hipMalloc(&out, M * N * 2));
    std::vector<std::pair<int, int>> forms;
    for (int ai = 5; ai < argc; ++ai) {
      int v = 4, st = 1;
      sscanf(argv[ai], "%d:%d", &v, &st);
      forms.push_back({v, st});
    }
    auto run = [&] {
      int rc = sglk_fp8_blockwise_scaled_mm(0, out, a, b, sa, sb, M, N, K, K, K, N, 1, M, 1, K / 128, SGLK_BF16);
      if (rc) { fprintf(stderr, "error: %s\n", sglk_last_error()); exit(1); }
    };
    std::vector<std::vector<float>> all(forms.size());
    const int rounds = getenv("ROUNDS") ? atoi(getenv("ROUNDS")) : 8;
    for (int r = 0; r < rounds; ++r)
      for (size_t f = 0; f < forms.size(); ++f) {
        sglk_debug_set_gemm_variant(forms[f].first);
        sglk_debug_set_gemm_stagger(forms[f].second);
        std::vector<float> ms;
        time_ms(run, r == 0 ? 300 : 40, 40, &ms);
        all[f].insert(all[f].end(), ms.begin(), ms.end());
      }
    for (size_t f = 0; f < forms.size(); ++f) {
      std::sort(all[f].begin(), all[f].end());
      const double med = all[f][all[f].size() / 2];
      printf("gemmab M=%lld N=%lld K=%lld variant=%d stagger=%d: median %.4f ms  min %.4f  p90 %.4f -> %.1f TFLOP/s\n", (long long)M,
             (long long)N, (long long)K, forms[f].first, forms[f].second, med, all[f][0], all[f][all[f].size() * 9 / 10],
             2.0 * M * N * K / med / 1e9);
    }
    return 0;
  }
  if (!strcmp(argv[1], "stamps")) {
    // in-kernel timeline of the persistent blockwise kernel (variant 18): per wave and K block the s_memtime
    // stamps t0 = compute of the block done, t1 = own LDS-DMA landed + LDS reads returned, t2 = barrier released
    const int64_t M = atoll(argv[2]), N = atoll(argv[3]), K = atoll(argv[4]);
    void* a = dev_random_bytes(M * K, 1, true);
    void* b = dev_random_bytes(N * K, 2, true);
    float* sa = dev_random_floats(M * (K / 128), 3, 1e-4f, 1e-3f);
    float* sb = dev_random_floats((K / 128) * ((N + 127) / 128), 4, 1e-4f, 1e-3f);
    void* out;
    HIP_CHECK(hipMalloc(&out, M * N * 2));
    const int nwg = 256, nw = nwg * 8;
    uint32_t* st;
    HIP_CHECK(hipMalloc(&st, (size_t)nw * 64 * 4));
    HIP_CHECK(hipMemset(st, 0, (size_t)nw * 64 * 4));
    sglk_debug_set_gemm_variant(18);
    sglk_debug_set_gemm_stamps(st);
    for (int i = 0; i < 300; ++i)
      sglk_fp8_blockwise_scaled_mm(0, out, a, b, sa, sb, M, N, K, K, K, N, 1, M, 1, K / 128, SGLK_BF16);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> h((size_t)nw * 64);
    HIP_CHECK(hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost));
    // s_memtime counts shader cycles (MI355X_MICROARCH.md, cycle constants)
    double sum_c = 0, sum_d = 0, sum_b = 0; long n = 0;
    for (int w = 0; w < nw; ++w) {
      const uint32_t* s = &h[(size_t)w * 64];
      if (s[0] == 0 && s[1] == 0) continue;
      for (int k = 1; k < 20; ++k) {
        sum_c += (double)(uint32_t)(s[3 * k] - s[3 * (k - 1) + 2]);
        sum_d += (double)(uint32_t)(s[3 * k + 1] - s[3 * k]);
        sum_b += (double)(uint32_t)(s[3 * k + 2] - s[3 * k + 1]);
        ++n;
      }
    }
    printf("stamps over %ld (wave, K block) samples, shader cycles: compute %.1f  own-wait %.1f  barrier %.1f  (block total %.1f)\n",
           n, sum_c / n, sum_d / n, sum_b / n, (sum_c + sum_d + sum_b) / n);
    for (int wg : {0, 1, 100}) {
      for (int w = 0; w < 8; ++w) {
        const uint32_t* s = &h[((size_t)wg * 8 + w) * 64];
        printf("wg %3d wave %d:", wg, w);
        for (int k = 1; k < 9; ++k)
          printf(" [c%u w%u b%u]", (uint32_t)(s[3 * k] - s[3 * (k - 1) + 2]), (uint32_t)(s[3 * k + 1] - s[3 * k]),
                 (uint32_t)(s[3 * k + 2] - s[3 * k + 1]));
        printf("\n");
      }
    }
    return 0;
  }
  if (!strcmp(argv[1], "scaledmm")) {
    const int64_t M = atoll(argv[2]), N = atoll(argv[3]), K = atoll(argv[4]);
    void* a = dev_random_bytes(M * K, 1, true);
    void* b = dev_random_bytes(N * K, 2, true);
    float* sa = dev_random_floats(M, 3, 1e-4f, 1e-3f);
    float* sb = dev_random_floats(N, 4, 1e-4f, 1e-3f);
    void* out;
    HIP_CHECK(hipMalloc(&out, M * N * 2));
    if (getenv("GEMM_VARIANT")) sglk_debug_set_gemm_variant(atoi(getenv("GEMM_VARIANT")));
    for (int dt : {SGLK_FP8_E4M3, SGLK_INT8}) {
      std::vector<float> all;
      auto run = [&] {
        int rc = sglk_scaled_mm(0, out, a, b, sa, sb, nullptr, M, N, K, K, K, N, dt, SGLK_BF16);
        if (rc) { fprintf(stderr, "error: %s\n", sglk_last_error()); exit(1); }
      };
      const double ms = time_ms(run, 100, 50, &all);
      printf("scaled_mm %s M=%lld N=%lld K=%lld median %.4f ms min %.4f -> %.1f T(FL)OP/s\n", dt == SGLK_INT8 ? "int8" : "fp8",
             (long long)M, (long long)N, (long long)K, ms, all[0], 2.0 * M * N * K / ms / 1e9);
    }
    return 0;
  }
  if (!strcmp(argv[1], "stream")) {  // kbench stream ROWS ROW_BYTES: DRAM access-pattern probe (see stream_probe_kernel)
    const int64_t rows = atoll(argv[2]), row_bytes = atoll(argv[3]);
    uint8_t* w;
    HIP_CHECK(hipMalloc(&w, rows * row_bytes));
    HIP_CHECK(hipMemset(w, 1, rows * row_bytes));
    uint32_t* out;
    HIP_CHECK(hipMalloc(&out, 64));
    const int blocks = (int)(rows / 128);
    for (int pat = 0; pat < 3; ++pat)
      for (int depth : {1, 2, 4}) {
        auto run = [&] {
#define GO(P, D) if (pat == P && depth == D) stream_probe_kernel<P, D><<<blocks, 256>>>(w, out, (int)row_bytes);
          GO(0, 1) GO(0, 2) GO(0, 4) GO(1, 1) GO(1, 2) GO(1, 4) GO(2, 1) GO(2, 2) GO(2, 4)
#undef GO
        };
        std::vector<float> all;
        const double ms = time_ms(run, 5, 20, &all);
        printf("stream rows=%lld row_bytes=%lld pattern=%d depth=%d: median %.1f us min %.1f -> %.0f GB/s\n", (long long)rows,
               (long long)row_bytes, pat, depth, ms * 1e3, all[0] * 1e3, rows * row_bytes / ms / 1e6);
      }
    for (int v = 0; v < 6; ++v) {
      auto run = [&] {
        if (v == 0) stream_probe3_kernel<4, 1><<<blocks, 256>>>(w, out, (int)row_bytes);
        if (v == 1) stream_probe3_kernel<4, 2><<<blocks, 256>>>(w, out, (int)row_bytes);
        if (v == 2) stream_probe3_kernel<4, 3><<<blocks, 256>>>(w, out, (int)row_bytes);
        if (v == 3) stream_probe3_kernel<8, 2><<<blocks, 256>>>(w, out, (int)row_bytes);
        if (v == 4) stream_probe3_kernel<16, 2><<<blocks, 256>>>(w, out, (int)row_bytes);
        if (v == 5) stream_probe3_kernel<4, 8><<<blocks, 256>>>(w, out, (int)row_bytes);
      };
      std::vector<float> all;
      const double ms = time_ms(run, 5, 20, &all);
      const char* names[] = {"depth 4, 1 WG/CU", "depth 4, 2 WG/CU", "depth 4, 3 WG/CU", "depth 8, 2 WG/CU", "depth 16, 2 WG/CU", "depth 4, 8 WG/CU"};
      printf("stream rows=%lld row_bytes=%lld pattern=3 (%s): median %.1f us min %.1f -> %.0f GB/s\n", (long long)rows,
             (long long)row_bytes, names[v], ms * 1e3, all[0] * 1e3, rows * row_bytes / ms / 1e6);
    }
    return 0;
  }
  if (!strcmp(argv[1], "w4a16")) {  // kbench w4a16 N K ROWS [probe:mt ...]: int4 group-128 grouped GEMM, 8 experts, uniform rows
    const int64_t E = 8, N = atoll(argv[2]), K = atoll(argv[3]);
    std::vector<int32_t> hr(E, (int32_t)atoll(argv[4]));  // ROWS: one number (uniform) or 8 comma-separated counts
    if (strchr(argv[4], ',')) {
      char* sp = strdup(argv[4]);
      int i = 0;
      for (char* tok = strtok(sp, ","); tok && i < E; tok = strtok(nullptr, ",")) hr[i++] = atoi(tok);
    }
    int64_t total = 0;
    for (int32_t r : hr) total += r;
    const int64_t rows = total / E;
    void* w = dev_random_bytes(E * N * K / 2, 1, false);
    std::vector<uint16_t> hs(E * N * (K / 128), 0x3c00 >> 3);  // bf16 ~0.0078
    for (auto& x : hs) x = 0x3c00;                               // bf16 2^-7
    void *sc, *act, *out;
    HIP_CHECK(hipMalloc(&sc, hs.size() * 2));
    HIP_CHECK(hipMemcpy(sc, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    std::vector<uint16_t> ha(total * K);
    std::mt19937 rng(5);
    for (auto& x : ha) x = (uint16_t)(0x3c00 | (rng() & 0x80ff));  // bf16 +-(2^-7 .. 2^-6)
    HIP_CHECK(hipMalloc(&act, ha.size() * 2));
    HIP_CHECK(hipMemcpy(act, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&out, total * N * 2));
    int32_t* dr;
    HIP_CHECK(hipMalloc(&dr, E * 4));
    HIP_CHECK(hipMemcpy(dr, hr.data(), E * 4, hipMemcpyHostToDevice));
    std::vector<std::pair<int, int>> vars;
    for (int i = 5; i < argc; ++i) {
      int pr = 0, mt = 0;
      sscanf(argv[i], "%d:%d", &pr, &mt);
      vars.push_back({pr, mt});
    }
    if (vars.empty()) vars.push_back({0, 0});
    for (auto [pr, mt] : vars) {
      sglk_debug_set_w4a16_probe(pr, mt);
      auto run = [&] {
        int rc = sglk_moe_grouped_mm_w4a16(0, out, act, w, sc, nullptr, nullptr, dr, total, E, N, K, 128, 1, SGLK_BF16);
        if (rc) { fprintf(stderr, "error: %s\n", sglk_last_error()); exit(1); }
      };
      std::vector<float> all;
      const double ms = time_ms(run, 30, 50, &all);
      printf("w4a16 N=%lld K=%lld rows=%lld probe=%d mt=%d: median %.1f us min %.1f -> weights %.0f GB/s\n", (long long)N,
             (long long)K, (long long)rows, pr, mt, ms * 1e3, all[0] * 1e3, E * N * K / 2 / ms / 1e6);
    }
    return 0;
  }
  if (!strcmp(argv[1], "mla")) {
    const int64_t B = atoll(argv[2]), S = atoll(argv[3]), H = atoll(argv[4]);
    const int64_t page = 64, npages = S / page;
    void* cache = dev_random_bytes((size_t)B * npages * page * 576 * 2, 5, false);
    // random bf16 bit patterns can be NaN/Inf: clear the top exponent bit instead -> |x| < 2
    {
      const size_t n = (size_t)B * npages * page * 576;
      std::vector<uint16_t> h(n);
      std::mt19937 rng(7);
      for (auto& x : h) x = (uint16_t)(rng() & 0xBFFF);
      HIP_CHECK(hipMemcpy(cache, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    std::vector<uint16_t> hq((size_t)B * H * 576);
    { std::mt19937 rng(8); for (auto& x : hq) x = (uint16_t)(rng() & 0xBFFF); }
    if (getenv("MLA_GAUSS")) {  // the reference benchmark's distribution: q = N(0,1) * MLA_GAUSS (100 there), cache = N(0,1)
      const float qs = (float)atof(getenv("MLA_GAUSS"));
      auto to_bf16 = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); };
      std::mt19937 rng(11);
      std::normal_distribution<float> nd(0.f, 1.f);
      for (auto& x : hq) x = to_bf16(nd(rng) * qs);
      const size_t n = (size_t)B * npages * page * 576;
      std::vector<uint16_t> h(n);
      for (auto& x : h) x = to_bf16(nd(rng));
      HIP_CHECK(hipMemcpy(cache, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    void *qn, *qp, *out;
    HIP_CHECK(hipMalloc(&qn, B * H * 512 * 2));
    HIP_CHECK(hipMalloc(&qp, B * H * 64 * 2));
    HIP_CHECK(hipMalloc(&out, B * H * 512 * 2));
    HIP_CHECK(hipMemcpy(qn, hq.data(), B * H * 512 * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(qp, hq.data() + B * H * 512, B * H * 64 * 2, hipMemcpyHostToDevice));
    std::vector<int32_t> table(B * npages), lens(B, (int32_t)S);
    { std::mt19937 rng(9); for (auto& x : table) x = rng() % (B * npages); }
    int32_t *dt, *dl;
    HIP_CHECK(hipMalloc(&dt, table.size() * 4));
    HIP_CHECK(hipMalloc(&dl, B * 4));
    HIP_CHECK(hipMemcpy(dt, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dl, lens.data(), B * 4, hipMemcpyHostToDevice));
    if (getenv("MLA_W")) sglk_debug_set_mla_waves_per_group(atoi(getenv("MLA_W")));
    if (getenv("MLA_PROBE")) sglk_debug_set_mla_probe(atoi(getenv("MLA_PROBE")));
    for (int ai = 5; ai < argc || ai == 5; ++ai) {
      const int64_t splits = ai < argc ? atoll(argv[ai]) : -1;
      const int64_t wsz = sglk_mla_decode_workspace_size(S, B, H, splits);
      void* ws = nullptr;
      if (wsz) HIP_CHECK(hipMalloc(&ws, wsz));
      auto run = [&] {
        int rc = sglk_flash_mla_decode(0, out, qn, qp, cache, dl, dt, ws, wsz, B, H, page, npages, H * 512, 512, H * 64, 64,
                                       page * 576, npages, 0.0417f, splits, SGLK_BF16);
        if (rc) { fprintf(stderr, "error: %s\n", sglk_last_error()); exit(1); }
      };
      std::vector<float> all;
      const double bytes = (double)B * H * 576 * 2 + (double)B * S * 576 * 2 + table.size() * 4 + B * 4 + (double)B * H * 512 * 2;
      if (getenv("MLA_STAMPS")) {  // in-kernel stamps of the rows128x kernel: mean cycles per tile (wave 0) in each segment
        std::vector<int> probes;
        { std::string e = getenv("MLA_STAMPS"); size_t pos = 0; while (pos < e.size()) { probes.push_back(atoi(e.c_str() + pos)); pos = e.find(',', pos); if (pos == std::string::npos) break; ++pos; } }
        for (int prb : probes) {
          sglk_debug_set_mla_variant(prb);  // 90 / 94: the round-4 loop (94: no DMA), 70 + probe: the round-5 loop
          for (int i = 0; i < 20; ++i) run();
          HIP_CHECK(hipDeviceSynchronize());
          std::vector<unsigned long long> st(16 * 4 * 4096);
          sglk_debug_get_mla_stamps(st.data(), (int)st.size());
          const int nw = (int)std::min<int64_t>(4096, B * (splits > 0 ? splits : 2) * 4);
          for (int w = 0; w < 4; ++w) {  // per wave: the barrier wait of one is the others' extra work
            double sum[14] = {0}, tiles = 0;
            for (int i = w; i < nw; i += 4) { for (int k = 0; k < 14; ++k) sum[k] += (double)st[i * 16 + k]; tiles += (double)st[i * 16 + 15]; }
            printf("variant %2d wave %d: landed-wait %.0f barrier %.0f DMA-setup %.0f QK+PV %.0f tail %.0f = %.0f per tile | per launch: prologue %.0f epilogue %.0f\n",
                   prb, w, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, (sum[3] + sum[4]) / tiles, sum[8] / tiles,
                   (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[8]) / tiles, sum[5] / (nw / 4), sum[6] / (nw / 4));
            if (w == 3)
              printf("          in-kernel clock %.0f MHz (%.0f cycles in %.1f us per workgroup)\n", sum[12] / sum[13] * 100.0,
                     sum[12] / (nw / 4), sum[13] / (nw / 4) / 100.0);
          }
        }
        sglk_debug_set_mla_variant(0);
        if (ws) HIP_CHECK(hipFree(ws));
        if (ai >= argc) break;
        continue;
      }
      if (getenv("MLA_TIME_VARIANTS")) {  // interleaved rounds of the listed variants (0 = the release configuration)
        std::vector<int> vs;
        { std::string e = getenv("MLA_TIME_VARIANTS"); size_t pos = 0; while (pos < e.size()) { vs.push_back(atoi(e.c_str() + pos)); pos = e.find(',', pos); if (pos == std::string::npos) break; ++pos; } }
        for (int round = 0; round < 3; ++round)
          for (int v : vs) {
            sglk_debug_set_mla_variant(v);
            const double m = time_ms(run, 60, 30, &all);
            printf("mla[variant %3d] B=%lld S=%lld H=%lld splits=%lld median %.4f ms min %.4f -> %.1f GB/s, %.1f TFLOP/s\n", v,
                   (long long)B, (long long)S, (long long)H, (long long)splits, m, all[0], bytes / m / 1e6,
                   2.0 * B * H * S * 1088 / m / 1e9);
          }
        sglk_debug_set_mla_variant(0);
        if (ws) HIP_CHECK(hipFree(ws));
        if (ai >= argc) break;
        continue;
      }
      if (getenv("MLA_AB")) {  // interleaved rounds of the round-4 tile loop (variant 50) and the current one
        for (int round = 0; round < 3; ++round)
          for (int v : {50, 0}) {
            sglk_debug_set_mla_variant(v);
            const double m = time_ms(run, 60, 30, &all);
            printf("mla[%s] B=%lld S=%lld H=%lld splits=%lld median %.4f ms min %.4f -> %.1f GB/s, %.1f TFLOP/s\n",
                   v == 50 ? "r4 loop" : "r5 loop", (long long)B, (long long)S, (long long)H, (long long)splits, m, all[0],
                   bytes / m / 1e6, 2.0 * B * H * S * 1088 / m / 1e9);
          }
        sglk_debug_set_mla_variant(0);
        if (ws) HIP_CHECK(hipFree(ws));
        if (ai >= argc) break;
        continue;
      }
      const double ms = time_ms(run, 100, 50, &all);
      printf("mla B=%lld S=%lld H=%lld splits=%lld median %.4f ms min %.4f -> %.1f GB/s, %.1f TFLOP/s\n", (long long)B,
             (long long)S, (long long)H, (long long)splits, ms, all[0], bytes / ms / 1e6,
             2.0 * B * H * S * 1088 / ms / 1e9);
      if (ws) HIP_CHECK(hipFree(ws));
      if (ai >= argc) break;
    }
    return 0;
  }
  return 1;
}
