// the quantisation steps either side of fp8_scaled_mm / W4A16 weight preparation.
//
//   sgl_per_token_quant_fp8  (reference src/sycl/per_token_quant_fp8.cpp:36, :96-125, :201): per row
//       scale = rowmax(|x|) / 448;  q = e4m3( clamp(x * (scale == 0 ? 0 : 1 / scale), +-448) );  output_s[row] = scale
//   sgl_per_tensor_quant_fp8 (reference src/sycl/per_tensor_quant_fp8.cpp:46-47, :58-105, :121-161):
//       dynamic: output_s[0] = max(output_s[0], max|x| / 448) (the caller zero-initialises it), then
//       q = e4m3( clamp(x * (1 / (output_s[0] + 1e-8)), +-448) );  static: the same with the scale given
//   awq_dequantize           (reference src/sycl/awq_dequantize.cpp:15-51, :98-123; tests/test_awq_dequant.py:13-62):
//       out[k][8 c + i] = T( (nibble_o(i)(qweight[k][c]) - nibble_o(i)(qzeros[k / g][c])) * scales[k / g][8 c + i] ),
//       o = (0, 4, 1, 5, 2, 6, 3, 7), g = K / scales.rows
// All three are HBM streams: 16-byte vector accesses wherever the row length allows, one pass over the input for the
// per-token kernel (the row stays in registers between the max and the quantisation when it fits).
#include "common.h"

namespace sglk {
namespace {

constexpr float kFp8Max = 448.0f;

// ---- per token ---------------------------------------------------------------------------------------------------
// TPR threads per row - one wave (four rows per workgroup) or the whole 256-thread workgroup; V elements per access (8, 4, 2 or 1:
// the widest that divides the row length keeps rows aligned). A lane keeps 8 vectors in registers: a wave covers 4096 elements
// without a second read of the row, a workgroup 16384. (Round 5, late - a size sweep: one wave per row took 10.5 - 12 us for 1 .. 256
// rows of 14336 elements, two passes of 28 serial loads, where the group quantiser takes 2.4 - 3.8: rows longer than a wave's cache,
// and calls with too few rows to fill the chip with waves, now get a workgroup per row.)
template <typename T, int V, int TPR>
__global__ __launch_bounds__(256) void per_token_quant_fp8_kernel(const T* __restrict__ x, uint8_t* __restrict__ q,
                                                                  float* __restrict__ s, int64_t rows, int64_t cols) {
  const int lane = TPR == 64 ? (threadIdx.x & 63) : threadIdx.x;
  const int64_t row = TPR == 64 ? (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) : (int64_t)blockIdx.x;
  if (row >= rows) return;
  const T* xr = x + row * cols;
  uint8_t* qr = q + row * cols;
  constexpr int kCache = 8;  // vectors kept in registers per lane
  const int64_t nvec = cols / V;
  Vec<T, V> cache[kCache];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < kCache; ++i) {
    const int64_t v = lane + TPR * i;
    if (v < nvec) {
      cache[i] = load_vec<T, V>(xr + v * V);
#pragma unroll
      for (int e = 0; e < V; ++e) mx = fmaxf(mx, fabsf((float)cache[i][e]));
    }
  }
  for (int64_t v = lane + TPR * kCache; v < nvec; v += TPR) {
    const Vec<T, V> t = load_vec<T, V>(xr + v * V);
#pragma unroll
    for (int e = 0; e < V; ++e) mx = fmaxf(mx, fabsf((float)t[e]));
  }
  mx = wave_max(mx);
  if constexpr (TPR == 256) {  // (the maximum of exact values: the order of the reduction cannot change it)
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  }
  const float scale = mx / kFp8Max;
  const float inv = scale == 0.f ? 0.f : 1.0f / scale;
  if (lane == 0) s[row] = scale;
  auto quant = [&](const Vec<T, V>& t, int64_t v) {
    uint8_t o[V];
#pragma unroll
    for (int e = 0; e < V; e += 2) {
      const float a = fmaxf(fminf((float)t[e] * inv, kFp8Max), -kFp8Max);
      const float b = (e + 1 < V) ? fmaxf(fminf((float)t[e + 1] * inv, kFp8Max), -kFp8Max) : 0.f;
      const int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
      o[e] = (uint8_t)p;
      if (e + 1 < V) o[e + 1] = (uint8_t)(p >> 8);
    }
    if constexpr (V == 8) *reinterpret_cast<uint2*>(qr + v * V) = *reinterpret_cast<const uint2*>(o);
    else if constexpr (V == 4) *reinterpret_cast<uint32_t*>(qr + v * V) = *reinterpret_cast<const uint32_t*>(o);
    else if constexpr (V == 2) *reinterpret_cast<uint16_t*>(qr + v * V) = *reinterpret_cast<const uint16_t*>(o);
    else qr[v] = o[0];
  };
#pragma unroll
  for (int i = 0; i < kCache; ++i) {
    const int64_t v = lane + TPR * i;
    if (v < nvec) quant(cache[i], v);
  }
  for (int64_t v = lane + TPR * kCache; v < nvec; v += TPR) quant(load_vec<T, V>(xr + v * V), v);
}

// ---- per tensor --------------------------------------------------------------------------------------------------
// Both passes: a workgroup walks 16-KiB chunks (1024 vectors of 8 elements), four 16-byte loads in flight per lane. The second
// pass walks the chunks in REVERSE order: what the first pass read last is what the caches still hold (a 32 MB activation
// fits the Infinity Cache whole, the XCDs' L2s hold its tail).
template <typename T>
__global__ __launch_bounds__(256) void per_tensor_absmax_kernel(const T* __restrict__ x, float* __restrict__ s, int64_t n) {
  const int64_t nvec = n / 8, nchunk = (nvec + 1023) / 1024;
  float mx = 0.f;
  for (int64_t c = blockIdx.x; c < nchunk; c += 2 * gridDim.x) {  // (two chunks per trip: eight loads in flight per lane)
    Vec<T, 8> t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t cc = j < 4 ? c : c + gridDim.x;
      const int64_t v = cc * 1024 + (j & 3) * 256 + threadIdx.x;
      t[j] = load_vec<T, 8>(x + (v < nvec ? v : nvec - 1) * 8);  // (past the end: the last vector again - it counts anyway)
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf((float)t[j][e]));
  }
  if (nvec == 0) mx = 0.f;
  for (int64_t i = nvec * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    mx = fmaxf(mx, fabsf((float)x[i]));
  mx = wave_max(mx);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    // non-negative floats order like their bit patterns: an integer atomic max raises the scale monotonically
    atomicMax(reinterpret_cast<int*>(s), __float_as_int(mx / kFp8Max));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void per_tensor_quant_fp8_kernel(const T* __restrict__ x, uint8_t* __restrict__ q,
                                                                   const float* __restrict__ s, int64_t n, int reverse) {
  const float inv = 1.0f / (s[0] + 1e-8f);
  const int64_t nvec = n / 8, nchunk = (nvec + 1023) / 1024;
  for (int64_t c0 = blockIdx.x; c0 < nchunk; c0 += gridDim.x) {
    const int64_t c = reverse ? nchunk - 1 - c0 : c0;
    Vec<T, 8> t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = c * 1024 + j * 256 + threadIdx.x;
      t[j] = load_vec<T, 8>(x + (v < nvec ? v : nvec - 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = c * 1024 + j * 256 + threadIdx.x;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(-kFp8Max, fminf((float)t[j][e] * inv, kFp8Max));
      uint2 o;
      o.x = pack4_e4m3(f[0], f[1], f[2], f[3]);
      o.y = pack4_e4m3(f[4], f[5], f[6], f[7]);
      if (v < nvec) *reinterpret_cast<uint2*>(q + v * 8) = o;
    }
  }
  for (int64_t i = nvec * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float f = fmaxf(-kFp8Max, fminf((float)x[i] * inv, kFp8Max));
    q[i] = (uint8_t)__builtin_amdgcn_cvt_pk_fp8_f32(f, 0.f, 0, false);
  }
}

// ---- AWQ ---------------------------------------------------------------------------------------------------------
// one thread per packed int32: 8 outputs (16 bytes)
template <typename T>
__global__ __launch_bounds__(256) void awq_dequantize_kernel(const int32_t* __restrict__ qw, const T* __restrict__ scales,
                                                             const int32_t* __restrict__ qz, T* __restrict__ out,
                                                             int64_t K, int64_t C, int64_t group) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= K * C) return;
  const int64_t k = idx / C, c = idx - k * C, gi = k / group;
  const uint32_t w = (uint32_t)qw[idx], z = (uint32_t)qz[gi * C + c];
  const Vec<T, 8> sc = load_vec<T, 8>(scales + (gi * C + c) * 8);
  Vec<T, 8> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int sh = 4 * (((i & 1) << 2) | (i >> 1));  // nibble order 0, 4, 1, 5, 2, 6, 3, 7
    const float d = (float)(int)((w >> sh) & 15u) - (float)(int)((z >> sh) & 15u);
    o[i] = (T)(d * (float)sc[i]);  // small integer x 16-bit float: exact in fp32, one rounding (== arithmetic in T)
  }
  store_vec<T, 8>(out + idx * 8, o);
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_per_token_quant_fp8(sglk_stream_t stream, void* output_q, float* output_s, const void* input,
                                        int64_t rows, int64_t cols, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(rows >= 0 && cols > 0, "sgl_per_token_quant_fp8: bad shape");
  if (rows == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  // one wave per row only for rows a wave's register cache covers AND enough of them to fill the chip with waves
  const bool wave_rows = rows >= 2048;
  const int v = (cols % 8 == 0 && (uintptr_t)input % 16 == 0 && (uintptr_t)output_q % 8 == 0) ? 8
                : (cols % 4 == 0 && (uintptr_t)input % 8 == 0 && (uintptr_t)output_q % 4 == 0) ? 4
                : (cols % 2 == 0 && (uintptr_t)input % 4 == 0 && (uintptr_t)output_q % 2 == 0) ? 2 : 1;
  const bool one_wave = wave_rows && cols / v <= 64 * 8;
  const unsigned grid = one_wave ? (unsigned)cdiv(rows, 4) : (unsigned)rows;
#define SGLK_PTQ_GO(V_)                                                                                               \
  if (one_wave) per_token_quant_fp8_kernel<T, V_, 64><<<grid, 256, 0, st>>>((const T*)input, (uint8_t*)output_q, output_s, rows, cols); \
  else per_token_quant_fp8_kernel<T, V_, 256><<<grid, 256, 0, st>>>((const T*)input, (uint8_t*)output_q, output_s, rows, cols);
  SGLK_DISPATCH_FLOAT(dtype, T, {
    if (v == 8) { SGLK_PTQ_GO(8) } else if (v == 4) { SGLK_PTQ_GO(4) } else if (v == 2) { SGLK_PTQ_GO(2) } else { SGLK_PTQ_GO(1) }
  });
#undef SGLK_PTQ_GO
  return check_launch("sgl_per_token_quant_fp8");
}

extern "C" int sglk_per_tensor_quant_fp8(sglk_stream_t stream, void* output_q, float* output_s, const void* input,
                                         int64_t numel, int is_static, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(numel >= 0, "sgl_per_tensor_quant_fp8: bad size");
  SGLK_REQUIRE((uintptr_t)input % 16 == 0 && (uintptr_t)output_q % 8 == 0,
               "sgl_per_tensor_quant_fp8: input must be 16-byte and output 8-byte aligned");
  if (numel == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = cdiv(cdiv(numel, 8), 1024);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  // (the absmax pass ends in ONE atomic max per workgroup on ONE word: ~12 ns each, they serialise - 2048 workgroups spent
  // 25 us there, more than both passes' bytes take; two workgroups per CU stream just as fast and leave 512 atomics that
  // overlap the stream)
  const int64_t ablocks = blocks < 2 * num_cus() ? blocks : 2 * num_cus();
  SGLK_DISPATCH_FLOAT(dtype, T, {
    if (!is_static) per_tensor_absmax_kernel<T><<<(unsigned)ablocks, 256, 0, st>>>((const T*)input, output_s, numel);
    per_tensor_quant_fp8_kernel<T><<<(unsigned)blocks, 256, 0, st>>>((const T*)input, (uint8_t*)output_q, output_s, numel,
                                                                     is_static ? 0 : 1);
  });
  return check_launch("sgl_per_tensor_quant_fp8");
}

extern "C" int sglk_awq_dequantize(sglk_stream_t stream, void* out, const int32_t* qweight, const void* scales,
                                   const int32_t* qzeros, int64_t K, int64_t C, int64_t group_size, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(K >= 0 && C > 0 && group_size > 0 && K % group_size == 0, "awq_dequantize: bad shape");
  SGLK_REQUIRE(dtype == SGLK_F16 || dtype == SGLK_BF16, "awq_dequantize: scales must be Half or BFloat16");
  SGLK_REQUIRE((uintptr_t)out % 16 == 0 && (uintptr_t)scales % 16 == 0, "awq_dequantize: out / scales must be 16-byte aligned");
  if (K == 0) return SGLK_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)cdiv(K * C, 256);
  if (dtype == SGLK_F16)
    awq_dequantize_kernel<f16><<<grid, 256, 0, st>>>(qweight, (const f16*)scales, qzeros, (f16*)out, K, C, group_size);
  else
    awq_dequantize_kernel<bf16><<<grid, 256, 0, st>>>(qweight, (const bf16*)scales, qzeros, (bf16*)out, K, C, group_size);
  return check_launch("awq_dequantize");
}
