// TORCH_LIBRARY registration of the sgl_kernel operator surface for ROCm.
//
// Replaces reference src/torch_extension_sycl.cc (registry) and the host halves
// of src/sycl/*.cpp: each op validates its tensors the way the reference's host
// launcher does, then hands raw pointers + sizes + the current HIP stream to
// the C-ABI in include/sglk.h. Schemas of ops the reference registers are kept
// character for character (they are the drop-in contract; cited per op);
// schemas of the ops it only declares are authored here from
// include/sgl_kernel_ops.h and python/sgl_kernel/gemm.py call order.
//
// Dispatch key: CUDA (PyTorch-ROCm exposes HIP devices as "cuda").
// Built as the CPython extension `sgl_kernel.common_ops` (limited API), the
// module name python/sgl_kernel/__init__.py:14 of the reference imports.
#include <Python.h>
#include <ATen/ATen.h>
#include <ATen/hip/HIPGeneratorImpl.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <optional>
#include <tuple>
#include <vector>

#include "sglk.h"

namespace {

using at::Tensor;

#define SGLK_CALL(expr)                                             \
  do {                                                              \
    const int sglk_rc_ = (expr);                                    \
    TORCH_CHECK(sglk_rc_ == 0, "sgl_kernel: ", sglk_last_error()); \
  } while (0)

#define CHECK_GPU(x) TORCH_CHECK((x).is_cuda(), #x " must be a GPU (cuda/hip) tensor")
#define CHECK_LAST_DIM_CONTIGUOUS(x) \
  TORCH_CHECK((x).dim() == 0 || (x).stride(-1) == 1, #x " must be contiguous at the last dimension")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")

sglk_stream_t stream_of(const Tensor& t) {
  return (sglk_stream_t)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

int dtype_code(at::ScalarType t, const char* what) {
  switch (t) {
    case at::kFloat: return SGLK_F32;
    case at::kHalf: return SGLK_F16;
    case at::kBFloat16: return SGLK_BF16;
    case at::kFloat8_e4m3fn: return SGLK_FP8_E4M3;
    case at::kFloat8_e5m2: return SGLK_FP8_E5M2;
    case at::kChar: return SGLK_INT8;
    case at::kByte: return SGLK_U8;
    case at::kInt: return SGLK_I32;
    case at::kLong: return SGLK_I64;
    default: TORCH_CHECK(false, what, ": unsupported dtype ", t);
  }
  return -1;
}

// ---- RMSNorm family (reference src/sycl/RMSNorm.cpp:793-905, Norm.h:18-47) ----------------

// RowStrides of reference RMSNorm.cpp:50-64
sglk_row_strides row_strides(const Tensor& t) {
  TORCH_CHECK(t.dim() == 2 || t.dim() == 3, "get_row_strides: expected a 2D or 3D tensor, got ", t.dim(), "D");
  if (t.dim() == 2) return {t.stride(0), 1, 0};
  const int64_t outer = t.stride(0), inner_size = t.size(1), inner_stride = t.stride(1);
  if (t.size(0) == 1 || outer == inner_size * inner_stride) return {inner_stride, 1, 0};
  return {outer, inner_size, inner_stride};
}

std::tuple<int64_t, int64_t> check_norm_inputs(const Tensor& input, const Tensor& weight) {
  CHECK_GPU(input);
  CHECK_LAST_DIM_CONTIGUOUS(input);
  TORCH_CHECK(input.dim() == 2 || input.dim() == 3, "input must be a 2D or 3D tensor");
  CHECK_LAST_DIM_CONTIGUOUS(weight);
  TORCH_CHECK(weight.device() == input.device(), "weight must be on the same device as input");
  TORCH_CHECK(weight.dim() == 1, "weight must be 1-D");
  TORCH_CHECK(input.size(-1) == weight.size(0), "weight size must equal the hidden size");
  const int64_t n = input.size(-1);
  TORCH_CHECK(n > 0, "hidden size must be positive");
  return {input.numel() / n, n};
}

void rmsnorm_impl(Tensor& output, const Tensor& input, const Tensor& weight, double eps, bool gemma) {
  auto [rows, n] = check_norm_inputs(input, weight);
  CHECK_GPU(output);
  CHECK_LAST_DIM_CONTIGUOUS(output);
  TORCH_CHECK(output.sizes() == input.sizes(), "output must have the shape of input");
  TORCH_CHECK(output.scalar_type() == input.scalar_type(), "output must have the dtype of input");
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_rmsnorm(stream_of(input), output.data_ptr(), input.data_ptr(), weight.data_ptr(), rows, n,
                         row_strides(input), row_strides(output), (float)eps,
                         dtype_code(input.scalar_type(), "rmsnorm"),
                         dtype_code(weight.scalar_type(), "rmsnorm weight"), gemma ? 1 : 0));
}

void fused_add_rmsnorm_impl(Tensor& input, Tensor& residual, const Tensor& weight, double eps, bool gemma,
                            const char* name) {
  TORCH_CHECK(input.is_contiguous(), name, ": input must be contiguous");
  TORCH_CHECK(residual.is_contiguous(), name, ": residual must be contiguous");
  auto [rows, n] = check_norm_inputs(input, weight);
  CHECK_GPU(residual);
  TORCH_CHECK(residual.sizes() == input.sizes(), name, ": residual must have the shape of input");
  TORCH_CHECK(residual.scalar_type() == input.scalar_type(), name, ": residual must have the dtype of input");
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_fused_add_rmsnorm(stream_of(input), input.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                   rows, n, (float)eps, dtype_code(input.scalar_type(), name),
                                   dtype_code(weight.scalar_type(), name), gemma ? 1 : 0));
}

void rmsnorm(Tensor& output, Tensor& input, Tensor& weight, double eps) {
  rmsnorm_impl(output, input, weight, eps, false);
}
void gemma_rmsnorm(Tensor& output, Tensor& input, Tensor& weight, double eps) {
  rmsnorm_impl(output, input, weight, eps, true);
}
void fused_add_rmsnorm(Tensor input, Tensor residual, Tensor weight, double eps) {
  fused_add_rmsnorm_impl(input, residual, weight, eps, false, "fused_add_rmsnorm");
}
void gemma_fused_add_rmsnorm(Tensor& input, Tensor& residual, Tensor& weight, double eps) {
  fused_add_rmsnorm_impl(input, residual, weight, eps, true, "gemma_fused_add_rmsnorm");
}

// ---- activation-and-mul (reference src/sycl/TripleOps.cpp:140-235) -------------------------

void act_and_mul_impl(Tensor& out, Tensor& input, int act, const char* name) {
  CHECK_GPU(input);
  CHECK_GPU(out);
  TORCH_CHECK(input.scalar_type() == at::kHalf || input.scalar_type() == at::kBFloat16 ||
                  input.scalar_type() == at::kFloat,
              name, ": input must be Half, BFloat16 or Float");
  TORCH_CHECK(out.scalar_type() == input.scalar_type(), name, ": out must have the dtype of input");
  TORCH_CHECK(input.dim() >= 1 && input.size(-1) % 2 == 0, name, ": last dimension of input must be even");
  TORCH_CHECK(out.is_contiguous(), name, ": out must be contiguous");
  const int64_t d = input.size(-1) / 2;
  TORCH_CHECK(out.dim() == input.dim() && out.size(-1) == d && out.numel() * 2 == input.numel(), name,
              ": out must be input's shape with the last dimension halved");
  // the reference makes the input contiguous (TripleOps.cpp:141); out is written in place
  const Tensor in_c = input.contiguous();
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_act_and_mul(stream_of(input), out.data_ptr(), in_c.data_ptr(), out.numel() / d, d,
                             dtype_code(input.scalar_type(), name), act));
}
void silu_and_mul(Tensor& out, Tensor& input) { act_and_mul_impl(out, input, SGLK_ACT_SILU, "silu_and_mul"); }
void gelu_tanh_and_mul(Tensor& out, Tensor& input) {
  act_and_mul_impl(out, input, SGLK_ACT_GELU_TANH, "gelu_tanh_and_mul");
}
void gelu_and_mul(Tensor& out, Tensor& input) { act_and_mul_impl(out, input, SGLK_ACT_GELU, "gelu_and_mul"); }

// reference src/sycl/SiluAndMulClamp.cpp:170-180 (same checks and messages)
void silu_and_mul_clamp(Tensor& out, Tensor& input, double swiglu_limit) {
  CHECK_GPU(input);
  CHECK_GPU(out);
  TORCH_CHECK(input.scalar_type() == at::kHalf || input.scalar_type() == at::kBFloat16,
              "silu_and_mul_clamp: input must be Half or BFloat16");
  TORCH_CHECK(out.scalar_type() == input.scalar_type(), "silu_and_mul_clamp: dtype mismatch");
  TORCH_CHECK(input.dim() >= 1 && input.size(-1) % 2 == 0, "silu_and_mul_clamp: input last dim must be even");
  TORCH_CHECK(out.numel() * 2 == input.numel(), "silu_and_mul_clamp: output numel must be half of input numel");
  TORCH_CHECK(swiglu_limit > 0.0, "silu_and_mul_clamp: swiglu_limit must be > 0");
  TORCH_CHECK(out.is_contiguous(), "silu_and_mul_clamp: out must be contiguous");
  const int64_t d = input.size(-1) / 2;
  const Tensor in_c = input.contiguous();
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_silu_and_mul_clamp(stream_of(input), out.data_ptr(), in_c.data_ptr(), out.numel() / d, d,
                                    dtype_code(input.scalar_type(), "silu_and_mul_clamp"), (float)swiglu_limit));
}

// reference src/sycl/SwigluAlphaLimit.cpp:153-175 (same checks and messages)
Tensor swiglu_gpt_oss_sigmoid_alpha(Tensor x, double alpha, double limit) {
  CHECK_GPU(x);
  TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kHalf || x.scalar_type() == at::kBFloat16,
              "Only float32, float16, and bfloat16 are supported");
  TORCH_CHECK(x.is_contiguous(), "x must be contiguous");
  TORCH_CHECK(x.dim() == 2, "x must be 2D [B, 2H]");
  TORCH_CHECK(x.size(1) % 2 == 0, "Last dim must be even");
  const int64_t rows = x.size(0), hidden = x.size(1) / 2;
  Tensor y = at::empty({rows, hidden}, x.options());
  if (rows == 0 || hidden == 0) return y;
  const c10::OptionalDeviceGuard guard(x.device());
  SGLK_CALL(sglk_swiglu_alpha_limit(stream_of(x), y.data_ptr(), x.data_ptr(), rows, hidden,
                                    dtype_code(x.scalar_type(), "swiglu_gpt_oss_sigmoid_alpha"), (float)alpha, (float)limit));
  return y;
}

// ---- per-token-group quant (reference src/sycl/per_token_group_quant_8bit.cpp:222-386) ------

void sgl_per_token_group_quant_8bit(Tensor input, Tensor output_q, Tensor output_s, int64_t group_size,
                                    double eps, double min_8bit, double max_8bit, bool scale_ue8m0) {
  CHECK_GPU(input);
  CHECK_GPU(output_q);
  CHECK_GPU(output_s);
  CHECK_CONTIGUOUS(input);
  CHECK_CONTIGUOUS(output_q);
  TORCH_CHECK(group_size > 0 && input.numel() % group_size == 0, "input.numel() must be divisible by group_size");
  TORCH_CHECK(output_s.dim() == 2, "output_s must be 2-D");
  TORCH_CHECK(input.dim() >= 1 && input.size(-1) % group_size == 0,
              "the hidden dimension must be divisible by group_size");
  TORCH_CHECK(output_q.numel() == input.numel(), "output_q must have input's number of elements");
  const auto in_t = input.scalar_type();
  TORCH_CHECK(in_t == at::kHalf || in_t == at::kBFloat16 || in_t == at::kFloat,
              "sgl_per_token_group_quant_8bit: input dtype must be Float16, BFloat16, or Float32, got ", in_t);
  const auto q_t = output_q.scalar_type();
  TORCH_CHECK(q_t == at::kChar || q_t == at::kFloat8_e4m3fn,
              "sgl_per_token_group_quant_8bit: output_q dtype must be Int8 or Float8_e4m3fn, got ", q_t);
  const int64_t k = input.size(-1);
  const int64_t rows = input.numel() / k;
  const int64_t groups_per_row = k / group_size;
  // layout detection of reference :259
  const bool column_major = output_s.stride(0) < output_s.stride(1);
  int kind = 0;
  int64_t s_row = 0, s_col = 0;
  if (!scale_ue8m0) {
    TORCH_CHECK(output_s.scalar_type() == at::kFloat, "output_s must be float32 unless scale_ue8m0");
    TORCH_CHECK(output_s.numel() >= rows * groups_per_row, "output_s is too small");
    if (column_major) {
      // reference writes element (row, g) at g * stride(1) + row
      s_row = 1;
      s_col = output_s.stride(1);
    } else {
      // reference writes element (row, g) at the flat group index
      s_row = groups_per_row;
      s_col = 1;
    }
  } else if (column_major) {
    TORCH_CHECK(output_s.element_size() == 4, "column-major ue8m0 scales must be packed 4 per 32-bit element");
    kind = 2;
    s_col = output_s.stride(1);
  } else {
    TORCH_CHECK(output_s.element_size() == 1, "row-major ue8m0 scales must be a uint8 tensor");
    TORCH_CHECK(output_s.numel() >= rows * groups_per_row, "output_s is too small");
    kind = 1;
  }
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_per_token_group_quant_8bit(stream_of(input), input.data_ptr(), output_q.data_ptr(),
                                            output_s.data_ptr(), rows, k, (int)group_size, (float)eps,
                                            (float)min_8bit, (float)max_8bit, dtype_code(in_t, "input"),
                                            dtype_code(q_t, "output_q"), kind, s_row, s_col));
}

// ---- fp8 blockwise GEMM (declared only: reference include/sgl_kernel_ops.h:581-586) ----------

Tensor fp8_blockwise_scaled_mm(const Tensor& mat_a, const Tensor& mat_b, const Tensor& scales_a,
                               const Tensor& scales_b, at::ScalarType out_dtype) {
  CHECK_GPU(mat_a);
  CHECK_GPU(mat_b);
  CHECK_GPU(scales_a);
  CHECK_GPU(scales_b);
  TORCH_CHECK(mat_a.dim() == 2 && mat_b.dim() == 2, "mat_a and mat_b must be 2-D");
  TORCH_CHECK(mat_a.stride(1) == 1, "mat_a must be a row major tensor");
  TORCH_CHECK(mat_b.stride(0) == 1, "mat_b must be a column major tensor");
  TORCH_CHECK(mat_a.size(1) == mat_b.size(0), "mat_a and mat_b shapes cannot be multiplied");
  TORCH_CHECK(mat_a.scalar_type() == at::kFloat8_e4m3fn, "mat_a must be Float8_e4m3fn");
  TORCH_CHECK(mat_b.scalar_type() == at::kFloat8_e4m3fn, "mat_b must be Float8_e4m3fn");
  TORCH_CHECK(out_dtype == at::kHalf || out_dtype == at::kBFloat16, "out_dtype must be Half or BFloat16");
  TORCH_CHECK(scales_a.scalar_type() == at::kFloat && scales_b.scalar_type() == at::kFloat,
              "scales_a and scales_b must be Float32");
  const int64_t M = mat_a.size(0), K = mat_a.size(1), N = mat_b.size(1);
  TORCH_CHECK(K % 128 == 0, "mat_a.size(1) must be a multiple of 128 (one scale per 128-deep block)");
  const int64_t kb = K / 128, nb = (N + 127) / 128;
  TORCH_CHECK(scales_a.dim() == 2 && scales_a.size(0) == M && scales_a.size(1) == kb,
              "scales_a must have shape [M, K/128]");
  TORCH_CHECK(scales_b.dim() == 2 && scales_b.size(0) == kb && scales_b.size(1) == nb,
              "scales_b must have shape [K/128, ceil(N/128)]");
  Tensor out = at::empty({M, N}, mat_a.options().dtype(out_dtype));
  if (M == 0) return out;
  const c10::OptionalDeviceGuard guard(mat_a.device());
  // few rows over a deep K: K-slice units with fp32 partial tiles in a scratch tensor (include/sglk.h; 0 bytes: not that shape)
  const int64_t ws_bytes = sglk_fp8_blockwise_scaled_mm_workspace_size(M, N, K);
  Tensor ws;
  if (ws_bytes > 0) ws = at::empty({ws_bytes}, mat_a.options().dtype(at::kByte));
  SGLK_CALL(sglk_fp8_blockwise_scaled_mm_ws(stream_of(mat_a), out.data_ptr(), mat_a.data_ptr(), mat_b.data_ptr(),
                                            scales_a.data_ptr<float>(), scales_b.data_ptr<float>(), M, N, K,
                                            mat_a.stride(0), mat_b.stride(1), out.stride(0), scales_a.stride(0),
                                            scales_a.stride(1), scales_b.stride(0), scales_b.stride(1),
                                            dtype_code(out_dtype, "out_dtype"), ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes));
  return out;
}

// ---- fp8 / int8 per-token x per-channel GEMM (declared only: reference
// include/sgl_kernel_ops.h:567-580; call order python/sgl_kernel/gemm.py:13-42) -----------------

Tensor scaled_mm_impl(const Tensor& mat_a, const Tensor& mat_b, const Tensor& scales_a, const Tensor& scales_b,
                      at::ScalarType out_dtype, const std::optional<Tensor>& bias, bool is_int8, const char* name) {
  CHECK_GPU(mat_a);
  CHECK_GPU(mat_b);
  CHECK_GPU(scales_a);
  CHECK_GPU(scales_b);
  TORCH_CHECK(mat_a.dim() == 2 && mat_b.dim() == 2, name, ": mat_a and mat_b must be 2-D");
  TORCH_CHECK(mat_a.stride(1) == 1, name, ": mat_a must be a row major tensor");
  TORCH_CHECK(mat_b.stride(0) == 1, name, ": mat_b must be a column major tensor");
  TORCH_CHECK(mat_a.size(1) == mat_b.size(0), name, ": mat_a and mat_b shapes cannot be multiplied");
  const auto want = is_int8 ? at::kChar : at::kFloat8_e4m3fn;
  TORCH_CHECK(mat_a.scalar_type() == want && mat_b.scalar_type() == want, name, ": mat_a and mat_b must be ", want);
  TORCH_CHECK(out_dtype == at::kHalf || out_dtype == at::kBFloat16, name, ": out_dtype must be Half or BFloat16");
  const int64_t M = mat_a.size(0), K = mat_a.size(1), N = mat_b.size(1);
  TORCH_CHECK(scales_a.numel() == M && scales_a.is_contiguous() && scales_a.scalar_type() == at::kFloat, name,
              ": scales_a must be a contiguous Float32 tensor of M elements");
  TORCH_CHECK(scales_b.numel() == N && scales_b.is_contiguous() && scales_b.scalar_type() == at::kFloat, name,
              ": scales_b must be a contiguous Float32 tensor of N elements");
  const void* bias_ptr = nullptr;
  if (bias.has_value()) {
    TORCH_CHECK(bias->numel() == N && bias->is_contiguous() && bias->scalar_type() == out_dtype, name,
                ": bias must be a contiguous tensor of N elements of out_dtype");
    CHECK_GPU(*bias);
    bias_ptr = bias->data_ptr();
  }
  Tensor out = at::empty({M, N}, mat_a.options().dtype(out_dtype));
  if (M == 0) return out;
  const c10::OptionalDeviceGuard guard(mat_a.device());
  // 129 .. 1024 rows over a deep K: K-slice units with raw accumulators in a scratch tensor (include/sglk.h; 0 bytes: not that shape)
  const int64_t ws_bytes = sglk_scaled_mm_workspace_size(M, N, K);
  Tensor ws;
  if (ws_bytes > 0) ws = at::empty({ws_bytes}, mat_a.options().dtype(at::kByte));
  SGLK_CALL(sglk_scaled_mm_ws(stream_of(mat_a), out.data_ptr(), mat_a.data_ptr(), mat_b.data_ptr(),
                              scales_a.data_ptr<float>(), scales_b.data_ptr<float>(), bias_ptr, M, N, K,
                              mat_a.stride(0), mat_b.stride(1), out.stride(0),
                              is_int8 ? SGLK_INT8 : SGLK_FP8_E4M3, dtype_code(out_dtype, "out_dtype"),
                              ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes));
  return out;
}

Tensor fp8_scaled_mm(const Tensor& mat_a, const Tensor& mat_b, const Tensor& scales_a, const Tensor& scales_b,
                     at::ScalarType out_dtype, const std::optional<Tensor>& bias) {
  return scaled_mm_impl(mat_a, mat_b, scales_a, scales_b, out_dtype, bias, false, "fp8_scaled_mm");
}
Tensor int8_scaled_mm(const Tensor& mat_a, const Tensor& mat_b, const Tensor& scales_a, const Tensor& scales_b,
                      at::ScalarType out_dtype, const std::optional<Tensor>& bias) {
  return scaled_mm_impl(mat_a, mat_b, scales_a, scales_b, out_dtype, bias, true, "int8_scaled_mm");
}

// ---- per-token / per-tensor fp8 quantisation, AWQ dequantisation (reference src/sycl/per_token_quant_fp8.cpp:201,
//      per_tensor_quant_fp8.cpp:161, awq_dequantize.cpp:98-123) -----------------------------------------------

void sgl_per_token_quant_fp8(Tensor input, Tensor output_q, Tensor output_s) {
  CHECK_GPU(input);
  CHECK_GPU(output_q);
  CHECK_GPU(output_s);
  CHECK_CONTIGUOUS(input);
  CHECK_CONTIGUOUS(output_q);
  CHECK_CONTIGUOUS(output_s);
  TORCH_CHECK(input.dim() >= 1, "sgl_per_token_quant_fp8: input must have at least one dimension");
  TORCH_CHECK(output_q.scalar_type() == at::kFloat8_e4m3fn && output_q.numel() == input.numel(),
              "sgl_per_token_quant_fp8: output_q must be float8_e4m3fn with the shape of input");
  const int64_t cols = input.size(-1), rows = cols ? input.numel() / cols : 0;
  TORCH_CHECK(output_s.scalar_type() == at::kFloat && output_s.numel() == rows,
              "sgl_per_token_quant_fp8: output_s must hold one float32 per token");
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_per_token_quant_fp8(stream_of(input), output_q.data_ptr(), output_s.data_ptr<float>(), input.data_ptr(),
                                     rows, cols, dtype_code(input.scalar_type(), "input")));
}

void sgl_per_tensor_quant_fp8(Tensor input, Tensor output_q, Tensor output_s, bool is_static) {
  CHECK_GPU(input);
  CHECK_GPU(output_q);
  CHECK_GPU(output_s);
  CHECK_CONTIGUOUS(input);
  CHECK_CONTIGUOUS(output_q);
  TORCH_CHECK(output_q.scalar_type() == at::kFloat8_e4m3fn && output_q.numel() == input.numel(),
              "sgl_per_tensor_quant_fp8: output_q must be float8_e4m3fn with the shape of input");
  TORCH_CHECK(output_s.scalar_type() == at::kFloat && output_s.numel() == 1, "sgl_per_tensor_quant_fp8: output_s must be one float32");
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_per_tensor_quant_fp8(stream_of(input), output_q.data_ptr(), output_s.data_ptr<float>(), input.data_ptr(),
                                      input.numel(), is_static ? 1 : 0, dtype_code(input.scalar_type(), "input")));
}

Tensor awq_dequantize(Tensor qweight, Tensor scales, Tensor qzeros) {
  CHECK_GPU(qweight);
  CHECK_GPU(scales);
  CHECK_GPU(qzeros);
  CHECK_CONTIGUOUS(qweight);
  CHECK_CONTIGUOUS(scales);
  CHECK_CONTIGUOUS(qzeros);
  TORCH_CHECK(qweight.dim() == 2 && scales.dim() == 2 && qzeros.dim() == 2, "awq_dequantize: 2-D tensors expected");
  TORCH_CHECK(qweight.scalar_type() == at::kInt && qzeros.scalar_type() == at::kInt, "awq_dequantize: qweight / qzeros must be int32");
  const int64_t K = qweight.size(0), C = qweight.size(1);
  TORCH_CHECK(scales.size(0) > 0 && K % scales.size(0) == 0 && scales.size(1) == C * 8 && qzeros.size(0) == scales.size(0) &&
                  qzeros.size(1) == C,
              "awq_dequantize: scales must be [K / group, 8 C] and qzeros [K / group, C]");
  Tensor out = at::empty({K, C * 8}, scales.options());
  const c10::OptionalDeviceGuard guard(qweight.device());
  SGLK_CALL(sglk_awq_dequantize(stream_of(qweight), out.data_ptr(), qweight.data_ptr<int32_t>(), scales.data_ptr(),
                                qzeros.data_ptr<int32_t>(), K, C, K / scales.size(0), dtype_code(scales.scalar_type(), "scales")));
  return out;
}

// ---- QServe W4A8 (reference include/sgl_kernel_ops.h:1132-1148; python/sgl_kernel/gemm.py:314-356) ----------

static void qserve_common(const char* op, const Tensor& in_feats, const Tensor& kernel, const Tensor& wscales,
                          const Tensor& ascales, Tensor& out_feats) {
  CHECK_GPU(in_feats);
  CHECK_GPU(kernel);
  CHECK_GPU(wscales);
  CHECK_GPU(ascales);
  CHECK_GPU(out_feats);
  TORCH_CHECK(in_feats.dim() == 2 && kernel.dim() == 2 && out_feats.dim() == 2, op, ": in_feats, kernel, out_feats must be 2-D");
  TORCH_CHECK(in_feats.scalar_type() == at::kChar && kernel.scalar_type() == at::kChar, op, ": in_feats and kernel must be int8");
  TORCH_CHECK(out_feats.scalar_type() == at::kHalf, op, ": only out dtype float16 is supported");
  TORCH_CHECK(wscales.scalar_type() == at::kHalf && ascales.scalar_type() == at::kHalf, op, ": wscales / ascales must be float16");
  const int64_t M = in_feats.size(0), K = in_feats.size(1), N = kernel.size(0);
  TORCH_CHECK(kernel.size(1) * 2 == K && kernel.is_contiguous(), op, ": kernel must be a contiguous [N, K/2] tensor");
  TORCH_CHECK(in_feats.stride(1) == 1 && out_feats.stride(1) == 1, op, ": in_feats / out_feats rows must be contiguous");
  TORCH_CHECK(out_feats.size(0) == M && out_feats.size(1) == N, op, ": out_feats must be [M, N]");
  TORCH_CHECK(wscales.numel() == N && wscales.is_contiguous() && ascales.numel() == M && ascales.is_contiguous(), op,
              ": wscales must hold N and ascales M contiguous values");
}

void qserve_w4a8_per_chn_gemm(const Tensor& in_feats, const Tensor& kernel, const Tensor& wscales, const Tensor& ascales,
                              const Tensor& w_szs, const Tensor& a_ssums, Tensor& out_feats) {
  qserve_common("qserve_w4a8_per_chn_gemm", in_feats, kernel, wscales, ascales, out_feats);
  CHECK_GPU(w_szs);
  CHECK_GPU(a_ssums);
  const int64_t M = in_feats.size(0), K = in_feats.size(1), N = kernel.size(0);
  TORCH_CHECK(w_szs.scalar_type() == at::kHalf && w_szs.numel() == N && w_szs.is_contiguous() &&
                  a_ssums.scalar_type() == at::kHalf && a_ssums.numel() == M && a_ssums.is_contiguous(),
              "qserve_w4a8_per_chn_gemm: w_szs [N] and a_ssums [M] must be contiguous float16");
  const c10::OptionalDeviceGuard guard(in_feats.device());
  SGLK_CALL(sglk_qserve_w4a8_per_chn_gemm(stream_of(in_feats), out_feats.data_ptr(), in_feats.data_ptr(), kernel.data_ptr(),
                                          wscales.data_ptr(), ascales.data_ptr(), w_szs.data_ptr(), a_ssums.data_ptr(), M, N,
                                          K, in_feats.stride(0), out_feats.stride(0)));
}

void qserve_w4a8_per_group_gemm(const Tensor& in_feats, const Tensor& kernel, const Tensor& zeros, const Tensor& scales_i8,
                                const Tensor& wscales, const Tensor& ascales, Tensor& out_feats) {
  qserve_common("qserve_w4a8_per_group_gemm", in_feats, kernel, wscales, ascales, out_feats);
  CHECK_GPU(zeros);
  CHECK_GPU(scales_i8);
  const int64_t M = in_feats.size(0), K = in_feats.size(1), N = kernel.size(0);
  TORCH_CHECK(zeros.scalar_type() == at::kChar && scales_i8.scalar_type() == at::kChar && zeros.is_contiguous() &&
                  scales_i8.is_contiguous() && zeros.numel() == (K / 128) * N && scales_i8.numel() == (K / 128) * N,
              "qserve_w4a8_per_group_gemm: zeros and scales_i8 must be contiguous int8 [K/128, N]");
  const c10::OptionalDeviceGuard guard(in_feats.device());
  SGLK_CALL(sglk_qserve_w4a8_per_group_gemm(stream_of(in_feats), out_feats.data_ptr(), in_feats.data_ptr(),
                                            kernel.data_ptr(), zeros.data_ptr(), scales_i8.data_ptr(), wscales.data_ptr(),
                                            ascales.data_ptr(), M, N, K, in_feats.stride(0), out_feats.stride(0)));
}

// ---- flash_mla_decode (reference src/sycl/mla_decode.cpp:135-175, :192-223) ------------------

void flash_mla_decode(Tensor& out, Tensor& q_nope, Tensor& q_pe, Tensor& kv_c_and_k_pe_cache, Tensor& seq_lens,
                      Tensor& page_table, Tensor& workspace, double sm_scale, int64_t num_kv_splits) {
  CHECK_GPU(out);
  CHECK_GPU(q_nope);
  CHECK_GPU(q_pe);
  CHECK_GPU(kv_c_and_k_pe_cache);
  CHECK_GPU(seq_lens);
  CHECK_GPU(page_table);
  TORCH_CHECK(q_nope.dim() == 3 && q_pe.dim() == 3 && kv_c_and_k_pe_cache.dim() == 3 && out.dim() == 3,
              "flash_mla_decode: q_nope, q_pe, kv cache and out must be 3-D");
  const int64_t B = q_nope.size(0), H = q_nope.size(1);
  TORCH_CHECK(q_nope.size(2) == 512 && q_pe.size(2) == 64 && kv_c_and_k_pe_cache.size(2) == 576,
              "flash_mla_decode: expects kv_lora_rank 512 and qk_rope_head_dim 64");
  TORCH_CHECK(q_pe.size(0) == B && q_pe.size(1) == H, "flash_mla_decode: q_nope / q_pe shape mismatch");
  TORCH_CHECK(out.size(0) == B && out.size(1) == H && out.size(2) == 512 && out.is_contiguous(),
              "flash_mla_decode: out must be a contiguous [B, H, 512] tensor");
  const auto dt = q_nope.scalar_type();
  TORCH_CHECK(dt == at::kHalf || dt == at::kBFloat16, "flash_mla_decode: dtype must be Half or BFloat16");
  TORCH_CHECK(q_pe.scalar_type() == dt && kv_c_and_k_pe_cache.scalar_type() == dt && out.scalar_type() == dt,
              "flash_mla_decode: q_nope, q_pe, kv cache and out must share one dtype");
  TORCH_CHECK(q_nope.stride(2) == 1 && q_pe.stride(2) == 1, "flash_mla_decode: q last dimension must be contiguous");
  TORCH_CHECK(kv_c_and_k_pe_cache.stride(2) == 1 && kv_c_and_k_pe_cache.stride(1) == 576,
              "flash_mla_decode: kv cache rows must be contiguous");
  TORCH_CHECK(seq_lens.scalar_type() == at::kInt && seq_lens.numel() == B && seq_lens.is_contiguous(),
              "flash_mla_decode: seq_lens must be a contiguous int32 [B] tensor");
  TORCH_CHECK(page_table.scalar_type() == at::kInt && page_table.dim() == 2 && page_table.size(0) == B &&
                  page_table.stride(1) == 1,
              "flash_mla_decode: page_table must be an int32 [B, n] tensor");
  const int64_t page = kv_c_and_k_pe_cache.size(1);
  TORCH_CHECK(page == 16 || page == 32 || page == 64 || page == 128, "Unsupported page size: ", page);
  int64_t ws_bytes = 0;
  void* ws_ptr = nullptr;
  if (workspace.defined() && workspace.numel() > 0) {
    CHECK_GPU(workspace);
    TORCH_CHECK(workspace.is_contiguous(), "flash_mla_decode: workspace must be contiguous");
    ws_bytes = workspace.numel() * workspace.element_size();
    ws_ptr = workspace.data_ptr();
  }
  const c10::OptionalDeviceGuard guard(q_nope.device());
  SGLK_CALL(sglk_flash_mla_decode(stream_of(q_nope), out.data_ptr(), q_nope.data_ptr(), q_pe.data_ptr(),
                                  kv_c_and_k_pe_cache.data_ptr(), seq_lens.data_ptr<int32_t>(),
                                  page_table.data_ptr<int32_t>(), ws_ptr, ws_bytes, B, H, page, page_table.size(1),
                                  q_nope.stride(0), q_nope.stride(1), q_pe.stride(0), q_pe.stride(1),
                                  kv_c_and_k_pe_cache.stride(0), page_table.stride(0), (float)sm_scale,
                                  num_kv_splits, dtype_code(dt, "flash_mla_decode")));
}

int64_t flash_mla_get_workspace_size(int64_t max_seq_len, int64_t num_batches, int64_t num_heads, int64_t page_size,
                                     int64_t num_kv_splits) {
  if (num_kv_splits < 1) {
    TORCH_CHECK(num_heads > 0, "num_heads must be > 0 when num_kv_splits is auto-selected");
    TORCH_CHECK(page_size == 16 || page_size == 32 || page_size == 64 || page_size == 128,
                "Unsupported page size: ", page_size);
  }
  return sglk_mla_decode_workspace_size(max_seq_len, num_batches, num_heads, num_kv_splits);
}

// ---- flash_mla_prefill (reference src/sycl/mla_prefill.cpp; schema torch_extension_sycl.cc:377-383) ----------

void flash_mla_prefill(Tensor& out, Tensor& q_nope, Tensor& q_pe, Tensor& kv_c_and_k_pe_cache, Tensor& cu_seqlens_q,
                       Tensor& seq_lens, int64_t max_seqlen_q, Tensor& page_table, Tensor& workspace, double sm_scale,
                       bool causal, int64_t num_kv_splits) {
  (void)workspace;
  (void)num_kv_splits;  // reserved by the reference as well (attention.py:178-179)
  CHECK_GPU(out);
  CHECK_GPU(q_nope);
  CHECK_GPU(q_pe);
  CHECK_GPU(kv_c_and_k_pe_cache);
  CHECK_GPU(cu_seqlens_q);
  CHECK_GPU(seq_lens);
  CHECK_GPU(page_table);
  TORCH_CHECK(q_nope.dim() == 3 && q_pe.dim() == 3 && kv_c_and_k_pe_cache.dim() == 3 && out.dim() == 3,
              "flash_mla_prefill: q_nope, q_pe, kv cache and out must be 3-D");
  const int64_t total_q = q_nope.size(0), H = q_nope.size(1);
  TORCH_CHECK(q_nope.size(2) == 512 && q_pe.size(2) == 64 && kv_c_and_k_pe_cache.size(2) == 576,
              "flash_mla_prefill: expects kv_lora_rank 512 and qk_rope_head_dim 64");
  TORCH_CHECK(q_pe.size(0) == total_q && q_pe.size(1) == H, "flash_mla_prefill: q_nope / q_pe shape mismatch");
  TORCH_CHECK(out.size(0) >= total_q && out.size(1) == H && out.size(2) == 512 && out.is_contiguous(),
              "flash_mla_prefill: out must be a contiguous [>= total_q, H, 512] tensor");
  const auto dt = q_nope.scalar_type();
  TORCH_CHECK(dt == at::kHalf || dt == at::kBFloat16, "flash_mla_prefill: dtype must be Half or BFloat16");
  TORCH_CHECK(q_pe.scalar_type() == dt && kv_c_and_k_pe_cache.scalar_type() == dt && out.scalar_type() == dt,
              "flash_mla_prefill: q_nope, q_pe, kv cache and out must share one dtype");
  TORCH_CHECK(q_nope.stride(2) == 1 && q_pe.stride(2) == 1, "flash_mla_prefill: q last dimension must be contiguous");
  TORCH_CHECK(kv_c_and_k_pe_cache.stride(2) == 1 && kv_c_and_k_pe_cache.stride(1) == 576,
              "flash_mla_prefill: kv cache rows must be contiguous");
  const int64_t B = seq_lens.numel();
  TORCH_CHECK(cu_seqlens_q.scalar_type() == at::kInt && cu_seqlens_q.numel() == B + 1 && cu_seqlens_q.is_contiguous(),
              "flash_mla_prefill: cu_seqlens_q must be a contiguous int32 [B + 1] tensor");
  TORCH_CHECK(seq_lens.scalar_type() == at::kInt && seq_lens.is_contiguous(),
              "flash_mla_prefill: seq_lens must be a contiguous int32 [B] tensor");
  TORCH_CHECK(page_table.scalar_type() == at::kInt && page_table.dim() == 2 && page_table.size(0) == B &&
                  page_table.stride(1) == 1,
              "flash_mla_prefill: page_table must be an int32 [B, n] tensor");
  const int64_t page = kv_c_and_k_pe_cache.size(1);
  TORCH_CHECK(page == 16 || page == 32 || page == 64 || page == 128, "Unsupported page size: ", page);
  const c10::OptionalDeviceGuard guard(q_nope.device());
  SGLK_CALL(sglk_flash_mla_prefill(stream_of(q_nope), out.data_ptr(), q_nope.data_ptr(), q_pe.data_ptr(),
                                   kv_c_and_k_pe_cache.data_ptr(), cu_seqlens_q.data_ptr<int32_t>(),
                                   seq_lens.data_ptr<int32_t>(), page_table.data_ptr<int32_t>(), B, max_seqlen_q, H, page,
                                   page_table.size(1), q_nope.stride(0), q_nope.stride(1), q_pe.stride(0),
                                   q_pe.stride(1), kv_c_and_k_pe_cache.stride(0), page_table.stride(0),
                                   (float)sm_scale, causal ? 1 : 0, dtype_code(dt, "flash_mla_prefill")));
}

int64_t flash_mla_prefill_get_workspace_size(int64_t max_seq_len, int64_t num_batches, int64_t num_heads,
                                             int64_t page_size, int64_t num_kv_splits) {
  return sglk_flash_mla_prefill_workspace_size(max_seq_len, num_batches, num_heads, page_size, num_kv_splits);
}

// ---- MoE (reference src/sycl/TopKSoftMax.cpp:584-644, MoEAlign.cpp:313-383, MoEPrepareInputs.cpp,
//           GroupGemmW4A16Xe20.cpp:92-283) ----------------------------------------------------------

void topk_softmax(Tensor& topk_weights, Tensor& topk_indices, Tensor& gating_output, bool renormalize) {
  CHECK_GPU(gating_output);
  CHECK_GPU(topk_weights);
  CHECK_GPU(topk_indices);
  TORCH_CHECK(gating_output.dim() == 2, "gating_output must be 2D tensor, but got ", gating_output.dim(), "D");
  const int64_t n_tokens = gating_output.size(0), n_experts = gating_output.size(1);
  TORCH_CHECK(n_experts <= 256, "n_experts only support up to 256, but got ", n_experts);
  TORCH_CHECK(topk_weights.scalar_type() == at::kFloat, "topk_weights should be Float");
  TORCH_CHECK(topk_indices.scalar_type() == at::kInt, "topk_indices should be Int");
  TORCH_CHECK(topk_weights.dim() == 2, "topk_weights must be 2D tensor, but got ", topk_weights.dim(), "D");
  TORCH_CHECK(topk_indices.dim() == 2, "topk_indices must be 2D tensor, but got ", topk_indices.dim(), "D");
  TORCH_CHECK(topk_weights.size(0) == n_tokens, "topk_weights.size(0) must equal n_tokens, but got ",
              topk_weights.size(0), " vs ", n_tokens);
  TORCH_CHECK(topk_indices.size(0) == n_tokens, "topk_indices.size(0) must equal n_tokens, but got ",
              topk_indices.size(0), " vs ", n_tokens);
  const int64_t n_topk = topk_weights.size(1);
  TORCH_CHECK(topk_indices.size(1) == n_topk, "topk_indices.size(1) must equal topk_weights.size(1), but got ",
              topk_indices.size(1), " vs ", n_topk);
  TORCH_CHECK(0 < n_topk && n_topk <= std::min<int64_t>(n_experts, 64),
              "n_topk must satisfy 0 < n_topk <= min(n_experts, 64), but got n_topk=", n_topk,
              " and n_experts=", n_experts);
  CHECK_CONTIGUOUS(gating_output);
  CHECK_CONTIGUOUS(topk_weights);
  CHECK_CONTIGUOUS(topk_indices);
  const auto dt = gating_output.scalar_type();
  TORCH_CHECK(dt == at::kHalf || dt == at::kBFloat16 || dt == at::kFloat, "gating_output must be a floating tensor");
  const c10::OptionalDeviceGuard guard(gating_output.device());
  SGLK_CALL(sglk_topk_softmax(stream_of(gating_output), topk_weights.data_ptr<float>(),
                              topk_indices.data_ptr<int32_t>(), gating_output.data_ptr(), n_tokens, n_experts, n_topk,
                              renormalize ? 1 : 0, dtype_code(dt, "gating_output")));
}

void moe_align_block_size(Tensor topk_ids, int64_t num_experts, int64_t block_size, Tensor sorted_token_ids,
                          Tensor experts_ids, Tensor num_tokens_post_pad, Tensor cumsum_buffer,
                          bool pad_sorted_token_ids) {
  CHECK_GPU(topk_ids);
  CHECK_GPU(sorted_token_ids);
  CHECK_GPU(experts_ids);
  CHECK_GPU(num_tokens_post_pad);
  CHECK_GPU(cumsum_buffer);
  CHECK_CONTIGUOUS(topk_ids);
  TORCH_CHECK(sorted_token_ids.scalar_type() == at::kInt && experts_ids.scalar_type() == at::kInt &&
                  num_tokens_post_pad.scalar_type() == at::kInt && cumsum_buffer.scalar_type() == at::kInt,
              "moe_align_block_size: output tensors must be int32");
  TORCH_CHECK(cumsum_buffer.numel() >= num_experts + 1, "moe_align_block_size: cumsum_buffer needs num_experts + 1 elements");
  const auto it = topk_ids.scalar_type();
  TORCH_CHECK(it == at::kInt || it == at::kLong, "moe_align_block_size: topk_ids must be int32 or int64");
  const c10::OptionalDeviceGuard guard(topk_ids.device());
  SGLK_CALL(sglk_moe_align_block_size(stream_of(topk_ids), topk_ids.data_ptr(), dtype_code(it, "topk_ids"),
                                      topk_ids.numel(), num_experts, block_size, sorted_token_ids.data_ptr<int32_t>(),
                                      experts_ids.data_ptr<int32_t>(), num_tokens_post_pad.data_ptr<int32_t>(),
                                      cumsum_buffer.data_ptr<int32_t>(), pad_sorted_token_ids ? 1 : 0));
}

void prepare_moe_input(const Tensor& topk_ids, Tensor& expert_offsets, const std::optional<Tensor>& blockscale_offsets,
                       Tensor& problem_sizes1, Tensor& problem_sizes2, Tensor& input_permutation,
                       Tensor& output_permutation, int64_t num_experts, int64_t n, int64_t k) {
  CHECK_GPU(topk_ids);
  const auto it = topk_ids.scalar_type();
  TORCH_CHECK(it == problem_sizes1.scalar_type(), "problem_sizes1 must have same type as topk_ids");
  TORCH_CHECK(it == expert_offsets.scalar_type(), "expert_offsets must have same type as topk_ids");
  TORCH_CHECK(it == problem_sizes2.scalar_type(), "problem_sizes2 must have same type as topk_ids");
  TORCH_CHECK(it == input_permutation.scalar_type(), "input_permutation must have same type as topk_ids");
  TORCH_CHECK(it == output_permutation.scalar_type(), "output_permutation must have same type as topk_ids");
  TORCH_CHECK(it == at::kInt || it == at::kLong, "prepare_moe_input: index tensors must be int32 or int64");
  TORCH_CHECK(!blockscale_offsets.has_value(), "prepare_moe_input: blockscale_offsets is not supported on this build");
  TORCH_CHECK(topk_ids.dim() == 2 && topk_ids.is_contiguous(), "prepare_moe_input: topk_ids must be contiguous [tokens, topk]");
  TORCH_CHECK(expert_offsets.numel() >= num_experts && problem_sizes1.numel() >= 3 * num_experts &&
                  problem_sizes2.numel() >= 3 * num_experts && input_permutation.numel() >= topk_ids.numel() &&
                  output_permutation.numel() >= topk_ids.numel(),
              "prepare_moe_input: an output tensor is too small");
  const c10::OptionalDeviceGuard guard(topk_ids.device());
  SGLK_CALL(sglk_prepare_moe_input(stream_of(topk_ids), topk_ids.data_ptr(), expert_offsets.data_ptr(),
                                   problem_sizes1.data_ptr(), problem_sizes2.data_ptr(), input_permutation.data_ptr(),
                                   output_permutation.data_ptr(), topk_ids.numel(), topk_ids.size(1), num_experts, n, k,
                                   dtype_code(it, "topk_ids")));
}

void scatter_tokens_to_experts(const Tensor& input, const Tensor& src2dst_map, Tensor& output) {
  CHECK_GPU(input);
  CHECK_GPU(src2dst_map);
  CHECK_GPU(output);
  TORCH_CHECK(input.scalar_type() == output.scalar_type(), "Input and output tensors must have the same data type");
  TORCH_CHECK(input.dim() == 2 && output.dim() == 2 && input.is_contiguous() && output.is_contiguous(),
              "scatter_tokens_to_experts: input and output must be contiguous 2-D tensors");
  TORCH_CHECK(input.size(1) == output.size(1), "scatter_tokens_to_experts: hidden sizes differ");
  TORCH_CHECK(src2dst_map.scalar_type() == at::kInt && src2dst_map.is_contiguous(),
              "scatter_tokens_to_experts: src2dst_map must be a contiguous int32 tensor");
  const int64_t tokens = input.size(0);
  if (tokens == 0) return;
  TORCH_CHECK(output.size(0) % tokens == 0, "scatter_tokens_to_experts: output rows must be a multiple of input rows");
  const int64_t topk = output.size(0) / tokens;
  TORCH_CHECK(src2dst_map.numel() >= tokens * topk, "scatter_tokens_to_experts: src2dst_map is too small");
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_scatter_tokens_to_experts(stream_of(input), input.data_ptr(), src2dst_map.data_ptr<int32_t>(),
                                           output.data_ptr(), tokens, topk, input.size(1) * input.element_size()));
}

void apply_shuffle_mul_sum(const Tensor& input, Tensor& output, const Tensor& permutation,
                           double routed_scaling_factor, const std::optional<Tensor>& factors) {
  CHECK_GPU(input);
  CHECK_GPU(output);
  CHECK_GPU(permutation);
  TORCH_CHECK(input.dim() == 2 && output.dim() == 2 && input.is_contiguous() && output.is_contiguous(),
              "apply_shuffle_mul_sum: input and output must be contiguous 2-D tensors");
  TORCH_CHECK(input.scalar_type() == output.scalar_type() && input.size(1) == output.size(1),
              "apply_shuffle_mul_sum: input and output must share dtype and hidden size");
  TORCH_CHECK(permutation.scalar_type() == at::kInt && permutation.is_contiguous(),
              "apply_shuffle_mul_sum: permutation must be a contiguous int32 tensor");
  const int64_t m = output.size(0);
  if (m == 0) return;
  const int64_t topk = permutation.numel() / m;
  const void* fptr = nullptr;
  int fdt = SGLK_F32;
  if (factors.has_value()) {
    CHECK_GPU(*factors);
    TORCH_CHECK(factors->is_contiguous() && factors->numel() >= m * topk, "apply_shuffle_mul_sum: bad factors tensor");
    fptr = factors->data_ptr();
    fdt = dtype_code(factors->scalar_type(), "factors");
  }
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_apply_shuffle_mul_sum(stream_of(input), input.data_ptr(), output.data_ptr(),
                                       permutation.data_ptr<int32_t>(), fptr, m, topk, output.size(1),
                                       (float)routed_scaling_factor, dtype_code(input.scalar_type(), "input"), fdt));
}

// split_ws != nullptr: the K-split form (sglk_moe_grouped_mm_w4a16_splitk); returns the row block of the split (0: not split)
static int64_t moe_w4a16_impl(Tensor& output, const Tensor& activations, const Tensor& packed_weights,
                          const Tensor& scales, const std::optional<Tensor>& zeros,
                          const std::optional<Tensor>& bias, const Tensor& rows_per_expert, int64_t n_experts,
                          bool is_int4, int64_t group_size, int64_t fused_act, double act_limit,
                          const std::optional<Tensor>& row_map = std::nullopt, Tensor* split_ws = nullptr, double act_alpha = 0.0) {
  CHECK_GPU(output);
  CHECK_GPU(activations);
  CHECK_GPU(packed_weights);
  CHECK_GPU(scales);
  CHECK_GPU(rows_per_expert);
  CHECK_CONTIGUOUS(output);
  CHECK_CONTIGUOUS(activations);
  CHECK_CONTIGUOUS(packed_weights);
  CHECK_CONTIGUOUS(scales);
  CHECK_CONTIGUOUS(rows_per_expert);
  TORCH_CHECK(output.dim() == 2, "output must be 2D [total_m, N]");
  TORCH_CHECK(activations.dim() == 2, "activations must be 2D [total_m, K]");
  TORCH_CHECK(rows_per_expert.dim() == 1, "rows_per_expert must be 1D [E]");
  const int32_t* map_ptr = nullptr;
  if (row_map.has_value()) {  // activations [tokens, K], row r of the grouped problem reads activations[row_map[r]]
    CHECK_GPU(*row_map);
    TORCH_CHECK(row_map->dim() == 1 && row_map->scalar_type() == at::kInt && row_map->is_contiguous(),
                "row_map must be a contiguous 1D int32 tensor [total_m]");
    map_ptr = row_map->data_ptr<int32_t>();
  }
  const int64_t total_m = map_ptr ? row_map->size(0) : activations.size(0), gemm_k = activations.size(1);
  TORCH_CHECK(packed_weights.dim() == 3, "packed_weights must be 3D [E, N, K/2]");
  const int64_t gemm_n = packed_weights.size(1);
  TORCH_CHECK(packed_weights.size(0) == n_experts, "packed_weights.size(0) must equal n_experts");
  TORCH_CHECK(packed_weights.size(2) == gemm_k / 2, "packed_weights.size(2) must equal K/2 (two 4-bit values per byte)");
  TORCH_CHECK(packed_weights.scalar_type() == at::kChar || packed_weights.scalar_type() == at::kByte,
              "packed_weights must be int8 or uint8");
  TORCH_CHECK(group_size == 32 || group_size == 64 || group_size == 128 || group_size == 256,
              "group_size must be 32, 64, 128 or 256; got ", group_size);
  TORCH_CHECK(gemm_k % group_size == 0, "K must be a multiple of group_size");
  TORCH_CHECK(scales.dim() == 3, "scales must be 3D [E, N, K/group_size]");
  TORCH_CHECK(scales.size(0) == n_experts, "scales.size(0) must equal n_experts");
  TORCH_CHECK(scales.size(1) == gemm_n, "scales.size(1) must equal N");
  TORCH_CHECK(scales.size(2) == gemm_k / group_size, "scales.size(2) must equal K/group_size");
  if (is_int4) {
    TORCH_CHECK(scales.scalar_type() == activations.scalar_type(), "int4 scales dtype must match activations dtype");
  } else {  // mxfp4 (reference GroupGemmW4A16Xe20.cpp:140-168): E8M0 scale bytes, one per 32 weights, no zero points
    TORCH_CHECK(scales.scalar_type() == at::kByte || scales.scalar_type() == at::kFloat8_e8m0fnu,
                "mxfp4 scales must be uint8 or float8_e8m0fnu (E8M0 bytes)");
    TORCH_CHECK(group_size == 32, "mxfp4 weights use group_size 32, got ", group_size);
    TORCH_CHECK(!zeros.has_value(), "mxfp4 weights have no zero points");
  }
  TORCH_CHECK(n_experts > 0, "n_experts must be positive");
  TORCH_CHECK(n_experts == rows_per_expert.size(0), "rows_per_expert must have n_experts elements");
  TORCH_CHECK(rows_per_expert.scalar_type() == at::kInt, "rows_per_expert must be int32");
  TORCH_CHECK(output.size(0) == total_m, map_ptr ? "output rows must match row_map's length" : "output rows must match activations rows");
  TORCH_CHECK(fused_act >= 0 && fused_act <= 5,
              "activation_type must be 0 (none), 1 (silu), 2 (gelu), 3 (relu2), 4 (clamped swiglu) or 5 (gpt-oss swiglu)");
  const bool gated = fused_act == 1 || fused_act == 2 || fused_act == 4 || fused_act == 5;
  TORCH_CHECK(output.size(1) == (gated ? gemm_n / 2 : gemm_n), gated ? "output must have N / 2 columns (gate rows, then up rows in W)"
                                                                     : "output must have N columns");
  TORCH_CHECK(gemm_n % 8 == 0, "N must be divisible by 8");
  TORCH_CHECK(activations.scalar_type() == at::kBFloat16 || activations.scalar_type() == at::kHalf,
              "activations must be bfloat16 or half");
  TORCH_CHECK(output.scalar_type() == activations.scalar_type(), "output dtype must match activations dtype");
  const float* bias_ptr = nullptr;
  if (bias.has_value()) {
    CHECK_GPU(*bias);
    TORCH_CHECK(bias->scalar_type() == at::kFloat, "bias must be float32");
    TORCH_CHECK(bias->dim() == 2 && bias->is_contiguous(), "bias must be 2D [E, N]");
    TORCH_CHECK(bias->size(0) == n_experts && bias->size(1) == gemm_n, "bias shape must be [E, N]");
    bias_ptr = bias->data_ptr<float>();
  }
  const void* zeros_ptr = nullptr;
  if (zeros.has_value()) {
    CHECK_GPU(*zeros);
    TORCH_CHECK(zeros->scalar_type() == scales.scalar_type(), "zeros dtype must match int4 scales dtype");
    TORCH_CHECK(zeros->sizes() == scales.sizes() && zeros->is_contiguous(),
                "zeros shape must match scales shape [E, N, K/group_size]");
    zeros_ptr = zeros->data_ptr();
  }
  const c10::OptionalDeviceGuard guard(activations.device());
  // the C-ABI wants 16-byte aligned scales / zeros (vector loads of a row's groups): a view at an odd storage offset is
  // copied once here instead of being refused
  Tensor scales_al = scales, zeros_al;
  if (reinterpret_cast<uintptr_t>(scales.data_ptr()) % 16 != 0) scales_al = scales.clone();
  if (zeros_ptr != nullptr && reinterpret_cast<uintptr_t>(zeros_ptr) % 16 != 0) {
    zeros_al = zeros->clone();
    zeros_ptr = zeros_al.data_ptr();
  }
  if (split_ws != nullptr) {
    CHECK_GPU(*split_ws);
    TORCH_CHECK(split_ws->scalar_type() == at::kFloat && split_ws->is_contiguous() && split_ws->numel() >= 2 * total_m * gemm_n,
                "moe_grouped_mm_nt_w4a16_splitk: ws must be a contiguous float32 tensor of at least 2 * total_m * N elements");
    TORCH_CHECK(!bias.has_value() && fused_act == 0 && map_ptr == nullptr, "moe_grouped_mm_nt_w4a16_splitk: no bias, activation or row map");
    int used = 0;
    SGLK_CALL(sglk_moe_grouped_mm_w4a16_splitk(stream_of(activations), output.data_ptr(), split_ws->data_ptr<float>(),
                                               activations.data_ptr(), packed_weights.data_ptr(), scales_al.data_ptr(), zeros_ptr,
                                               rows_per_expert.data_ptr<int32_t>(), total_m, n_experts, gemm_n, gemm_k, group_size,
                                               is_int4 ? 1 : 0, dtype_code(activations.scalar_type(), "activations"), &used));
    return used;
  }
  if (fused_act == 5) {
    SGLK_CALL(sglk_moe_grouped_mm_w4a16_swiglu(stream_of(activations), output.data_ptr(), activations.data_ptr(),
                                               packed_weights.data_ptr(), scales_al.data_ptr(), zeros_ptr, bias_ptr,
                                               rows_per_expert.data_ptr<int32_t>(), total_m, n_experts, gemm_n, gemm_k, group_size,
                                               is_int4 ? 1 : 0, dtype_code(activations.scalar_type(), "activations"),
                                               (float)act_alpha, (float)act_limit, map_ptr, activations.size(0)));
    return 0;
  }
  SGLK_CALL(sglk_moe_grouped_mm_w4a16_act(stream_of(activations), output.data_ptr(), activations.data_ptr(),
                                          packed_weights.data_ptr(), scales_al.data_ptr(), zeros_ptr, bias_ptr,
                                          rows_per_expert.data_ptr<int32_t>(), total_m, n_experts, gemm_n, gemm_k,
                                          group_size, is_int4 ? 1 : 0,
                                          dtype_code(activations.scalar_type(), "activations"), (int)fused_act, (float)act_limit,
                                          map_ptr, activations.size(0)));
  return 0;
}

// authored (no reference op): the down projection of fused_experts with the K range of its tiles split in two where that
// projection has fewer tiles than the GPU has CUs (include/sglk.h: sglk_moe_grouped_mm_w4a16_splitk). Returns the row block
// (128 / 256) when the split was used - the result is then in ws (two fp32 partial sums per row) except for the experts'
// remainders of up to half a block, which are in output, and apply_shuffle_mul_sum_splitk reads both - and 0 when output holds
// the whole product.
int64_t moe_grouped_mm_nt_w4a16_splitk(Tensor& output, Tensor& ws, const Tensor& activations, const Tensor& packed_weights,
                                    const Tensor& scales, const std::optional<Tensor>& zeros, const Tensor& rows_per_expert,
                                    int64_t n_experts, bool is_int4, int64_t group_size) {
  return moe_w4a16_impl(output, activations, packed_weights, scales, zeros, std::nullopt, rows_per_expert, n_experts, is_int4,
                        group_size, 0, 0.0, std::nullopt, &ws);
}

// host-only: the row block moe_grouped_mm_nt_w4a16_splitk would split this shape with, 0 = not (fused_experts allocates ws only then)
int64_t moe_w4a16_splitk_applies(int64_t total_m, int64_t n_experts, int64_t n, int64_t k, int64_t group_size, bool is_int4,
                              bool is_bf16) {
  return sglk_moe_w4a16_splitk_applies(total_m, n_experts, n, k, group_size, is_int4 ? 1 : 0, is_bf16 ? SGLK_BF16 : SGLK_F16);
}

void apply_shuffle_mul_sum_splitk(const Tensor& y, const Tensor& ws, Tensor& output, const Tensor& permutation,
                                  const Tensor& rows_per_expert, int64_t block_rows, double routed_scaling_factor,
                                  const std::optional<Tensor>& factors) {
  CHECK_GPU(y);
  CHECK_GPU(ws);
  CHECK_GPU(output);
  CHECK_GPU(permutation);
  CHECK_GPU(rows_per_expert);
  TORCH_CHECK(y.dim() == 2 && output.dim() == 2 && y.is_contiguous() && output.is_contiguous(),
              "apply_shuffle_mul_sum_splitk: y and output must be contiguous 2-D tensors");
  TORCH_CHECK(y.scalar_type() == output.scalar_type() && y.size(1) == output.size(1),
              "apply_shuffle_mul_sum_splitk: y and output must share dtype and hidden size");
  TORCH_CHECK(ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.numel() >= 2 * y.numel(),
              "apply_shuffle_mul_sum_splitk: ws must be a contiguous float32 tensor [2, rows, hidden]");
  TORCH_CHECK(permutation.scalar_type() == at::kInt && permutation.is_contiguous(),
              "apply_shuffle_mul_sum_splitk: permutation must be a contiguous int32 tensor");
  TORCH_CHECK(rows_per_expert.scalar_type() == at::kInt && rows_per_expert.is_contiguous() && rows_per_expert.dim() == 1,
              "apply_shuffle_mul_sum_splitk: rows_per_expert must be a contiguous 1-D int32 tensor");
  const int64_t m = output.size(0);
  if (m == 0) return;
  const int64_t topk = permutation.numel() / m;
  const void* fptr = nullptr;
  int fdt = SGLK_F32;
  if (factors.has_value()) {
    CHECK_GPU(*factors);
    TORCH_CHECK(factors->is_contiguous() && factors->numel() >= m * topk, "apply_shuffle_mul_sum_splitk: bad factors tensor");
    fptr = factors->data_ptr();
    fdt = dtype_code(factors->scalar_type(), "factors");
  }
  const c10::OptionalDeviceGuard guard(y.device());
  SGLK_CALL(sglk_apply_shuffle_mul_sum_splitk(stream_of(y), y.data_ptr(), ws.data_ptr<float>(), output.data_ptr(),
                                              permutation.data_ptr<int32_t>(), fptr, rows_per_expert.data_ptr<int32_t>(),
                                              rows_per_expert.numel(), block_rows, y.size(0), m, topk, output.size(1),
                                              (float)routed_scaling_factor, dtype_code(y.scalar_type(), "y"), fdt));
}

void moe_grouped_mm_nt_xe20_w4a16(Tensor& output, const Tensor& activations, const Tensor& packed_weights,
                                  const Tensor& scales, const std::optional<Tensor>& zeros,
                                  const std::optional<Tensor>& bias, const Tensor& rows_per_expert, int64_t n_experts,
                                  bool is_int4, int64_t group_size) {
  moe_w4a16_impl(output, activations, packed_weights, scales, zeros, bias, rows_per_expert, n_experts, is_int4, group_size, 0, 0.0);
}

// authored (no reference op: the reference runs GEMM 1 and the gate / up activation as two launches,
// python/sgl_kernel/moe.py:751-835): the same GEMM with the activation on its fp32 accumulators.
// activation_type: 1 silu, 2 gelu (tanh), 4 clamped swiglu (act_limit) - output [total_m, N / 2]; 3 relu2 - output [total_m, N];
// 5 the gpt-oss swiglu (act_alpha, act_limit; gate / up rows INTERLEAVED in the weights as the reference's fused 16-bit GEMM takes
// them, kernels/moe/xe20/bf16/moe_kernel.hpp:109-125) - output [total_m, N / 2]
// row_map (optional, int32 [total_m]): activations are the tokens [T, K] and row r reads activations[row_map[r]] - the
// reference's shuffle_rows (python/sgl_kernel/moe.py:739) folded into the GEMM's staging loads
void moe_grouped_mm_nt_w4a16_act(Tensor& output, const Tensor& activations, const Tensor& packed_weights,
                                 const Tensor& scales, const std::optional<Tensor>& zeros,
                                 const std::optional<Tensor>& bias, const Tensor& rows_per_expert, int64_t n_experts,
                                 bool is_int4, int64_t group_size, int64_t activation_type, double act_limit,
                                 const std::optional<Tensor>& row_map, double act_alpha) {
  moe_w4a16_impl(output, activations, packed_weights, scales, zeros, bias, rows_per_expert, n_experts, is_int4, group_size,
                 activation_type, act_limit, row_map, nullptr, act_alpha);
}

// ---- moe_grouped_mm_nt_xe20 (reference src/sycl/GroupGemmXe20.cpp:160-275) ---------------------------------

void moe_grouped_mm_nt_xe20(Tensor& output, const Tensor& activations, const Tensor& weights, const std::optional<Tensor>& bias,
                            const Tensor& total_rows_for_experts, int64_t n_experts, int64_t activation_type, bool fuse_act,
                            double gemm1_alpha, double gemm1_limit) {
  CHECK_GPU(output);
  CHECK_GPU(activations);
  CHECK_GPU(weights);
  CHECK_GPU(total_rows_for_experts);
  CHECK_CONTIGUOUS(activations);
  CHECK_CONTIGUOUS(output);
  TORCH_CHECK(activations.dim() == 2 && weights.dim() == 3, "activations must be 2D and weights 3D");
  const int64_t total_m = activations.size(0), gemm_k = activations.size(1), gemm_n = weights.size(1);
  TORCH_CHECK(weights.size(0) == n_experts, "weights must have n_experts as the first dimension");
  TORCH_CHECK(weights.size(2) == gemm_k && weights.stride(2) == 1, "weights must be [n_experts, N, K] with contiguous K");
  TORCH_CHECK(total_rows_for_experts.size(0) == n_experts && total_rows_for_experts.scalar_type() == at::kInt &&
                  total_rows_for_experts.is_contiguous(),
              "rows_for_experts must be an int32 tensor with one entry per expert");
  TORCH_CHECK(output.size(0) == total_m, "output must have the same number of rows as activations");
  TORCH_CHECK(activations.scalar_type() == weights.scalar_type() && output.scalar_type() == activations.scalar_type(),
              "activations, weights and output must have the same data type");
  TORCH_CHECK(activations.scalar_type() == at::kBFloat16 || activations.scalar_type() == at::kHalf,
              "Only bfloat16 and half are supported in moe_grouped_mm_nt");
  TORCH_CHECK(activation_type >= 0 && activation_type <= 3, "Unsupported activation_type: ", activation_type,
              ". Supported values are 0 (silu), 1 (gelu), 2 (swiglu_gpt_oss), 3 (relu2)");
  const float* bias_ptr = nullptr;
  if (bias.has_value()) {
    CHECK_GPU((*bias));
    TORCH_CHECK(bias->scalar_type() == at::kFloat, "moe_grouped_mm_nt_xe20: bias must be float32 (at::kFloat) to match kernel expectations");
    TORCH_CHECK(bias->dim() == 2 && bias->size(0) == n_experts && bias->size(1) == gemm_n && bias->is_contiguous(),
                "bias must be 2D [n_experts, N]");
    bias_ptr = bias->data_ptr<float>();
  }
  const c10::OptionalDeviceGuard guard(activations.device());
  const int dt = dtype_code(activations.scalar_type(), "activations");
  // fused epilogue codes of the C-ABI: 0 none, 1 silu (gated), 2 gelu (gated), 3 relu2; the gpt-oss swiglu (activation_type 2:
  // gate / up rows interleaved, reference moe_kernel.hpp:109-125) has its own entry point with alpha and limit
  int fused = 0;
  if (fuse_act) fused = activation_type == 0 ? 1 : activation_type == 1 ? 2 : activation_type == 2 ? 5 : 3;
  if (fused == 5) {
    TORCH_CHECK(gemm_n % 2 == 0 && output.size(1) == gemm_n / 2, "output must have half the number of columns as activations");
    TORCH_CHECK(gemm1_limit > 0.0, "moe_grouped_mm_nt_xe20: gemm1_limit must be positive");
    SGLK_CALL(sglk_moe_grouped_mm_swiglu(stream_of(activations), output.data_ptr(), activations.data_ptr(), weights.data_ptr(),
                                         bias_ptr, total_rows_for_experts.data_ptr<int32_t>(), total_m, n_experts, gemm_n, gemm_k,
                                         weights.stride(1), weights.stride(0), dt, (float)gemm1_alpha, (float)gemm1_limit));
    return;
  }
  if (fused == 1 || fused == 2) {
    TORCH_CHECK(gemm_n % 2 == 0 && output.size(1) == gemm_n / 2, "output must have half the number of columns as activations");
  } else {
    TORCH_CHECK(output.size(1) == gemm_n, "output must have the same number of columns as the weights have rows");
  }
  SGLK_CALL(sglk_moe_grouped_mm(stream_of(activations), output.data_ptr(), activations.data_ptr(), weights.data_ptr(), bias_ptr,
                                total_rows_for_experts.data_ptr<int32_t>(), total_m, n_experts, gemm_n, gemm_k,
                                weights.stride(1), weights.stride(0), dt, fused));
}

// ---- fwd / mha_fwd (reference src/sycl/flash_attention.cpp:1332-1435; the int/float narrowing the reference
//      does with make_pytorch_shim, include/sgl_kernel_torch_shim.h:94-122, is done inline here) --------------

std::tuple<Tensor, Tensor, Tensor, Tensor> mha_fwd(
    const Tensor& q, const Tensor& k, const Tensor& v, const std::optional<Tensor>& q_v, const Tensor& cu_seqlens_q,
    const Tensor& cu_seqlens_k, int64_t max_seqlen_q, int64_t max_seqlen_k, const std::optional<Tensor>& page_table,
    const std::optional<Tensor>& kv_batch_idx, const std::optional<Tensor>& leftpad_k,
    const std::optional<Tensor>& rotary_cos, const std::optional<Tensor>& rotary_sin,
    const std::optional<Tensor>& seqlens_rotary, const std::optional<Tensor>& q_descale,
    const std::optional<Tensor>& k_descale, const std::optional<Tensor>& v_descale, double softmax_scale,
    const std::optional<Tensor>& sinks, bool is_causal, int64_t window_size_left, int64_t window_size_right,
    double softcap, bool is_rotary_interleaved, const std::optional<Tensor>& scheduler_metadata,
    int64_t num_kv_splits, std::optional<bool> pack_gqa, int64_t sm_margin, const std::optional<Tensor>& out_) {
  CHECK_GPU(q);
  CHECK_GPU(k);
  CHECK_GPU(v);
  TORCH_CHECK(q.dim() == 3, "query must be in ragged format (total_q, h, d)");
  const auto q_type = q.scalar_type();
  TORCH_CHECK(q_type == at::kHalf || q_type == at::kBFloat16, "mha_fwd only supports Half and BFloat16, got", q_type);
  // fp8 KV cache (reference flash_attention.cpp:315-320, :561-572): K/V stored as e4m3fn / e5m2, one float descale each
  const bool fp8_kv = k.scalar_type() == at::kFloat8_e4m3fn || k.scalar_type() == at::kFloat8_e5m2;
  if (fp8_kv) {
    TORCH_CHECK(v.scalar_type() == k.scalar_type(), "key and value must have the same dtype");
    TORCH_CHECK(k_descale.has_value() && v_descale.has_value(), "fp8 KV cache requires k_descale and v_descale");
    TORCH_CHECK(k.dim() == 4, "fwd: the fp8 KV cache path needs a KV cache (paged or one row per slot)");
  } else {
    TORCH_CHECK(k.scalar_type() == q_type, "query and key must have the same dtype");
    TORCH_CHECK(v.scalar_type() == q_type, "query and value must have the same dtype");
  }
  CHECK_LAST_DIM_CONTIGUOUS(q);
  CHECK_LAST_DIM_CONTIGUOUS(k);
  CHECK_LAST_DIM_CONTIGUOUS(v);
  TORCH_CHECK(!q_v.has_value(), "q_v is not supported yet");  // as the reference: flash_attention.cpp:603-609
  TORCH_CHECK(!rotary_cos.has_value() && !rotary_sin.has_value() && !seqlens_rotary.has_value(),
              "fwd: in-kernel rotary embedding is not supported");
  // q_descale: the reference takes it and never reads it (q must be Half / BFloat16, flash_attention.cpp:308-312, :914-918;
  // only k_descale / v_descale reach its kernels, :561-572, :1092-1096): accepted and ignored here as well.
  if (q_descale.has_value()) {
    CHECK_GPU(*q_descale);
    TORCH_CHECK(q_descale->scalar_type() == at::kFloat, "q_descale must be float32");
  }
  // per-tensor descale: a scalar or an expanded scalar (reference get_per_tensor_descale_ptr, flash_attention.cpp:45-70)
  auto descale_ptr = [&](const std::optional<Tensor>& t, const char* name) -> const float* {
    if (!fp8_kv || !t.has_value()) return nullptr;
    TORCH_CHECK(t->scalar_type() == at::kFloat, name, " must be float32");
    TORCH_CHECK(t->device() == k.device(), name, " must be on the same device as the tensor it descales");
    TORCH_CHECK(t->numel() > 0, name, " must not be empty");
    bool scalar = t->numel() == 1;
    if (!scalar) {
      scalar = true;
      for (int64_t dim = 0; dim < t->dim(); ++dim)
        if (t->size(dim) > 1 && t->stride(dim) != 0) scalar = false;
    }
    TORCH_CHECK(scalar, name, " uses a per-tensor descale: pass a single value (or an expanded view of one)");
    return t->data_ptr<float>();
  };
  const float* k_descale_ptr = descale_ptr(k_descale, "k_descale");
  const float* v_descale_ptr = descale_ptr(v_descale, "v_descale");
  TORCH_CHECK(cu_seqlens_q.scalar_type() == at::kInt && cu_seqlens_q.is_contiguous() && cu_seqlens_q.is_cuda(),
              "cu_seqlens_q must have dtype torch.int32");
  TORCH_CHECK(cu_seqlens_k.scalar_type() == at::kInt && cu_seqlens_k.is_contiguous() && cu_seqlens_k.is_cuda(),
              "cu_seqlens_k must have dtype torch.int32");
  const int64_t batch = cu_seqlens_q.size(0) - 1;
  const int64_t total_q = q.size(0), num_heads = q.size(1), head_size = q.size(2);
  const int64_t num_heads_k = k.size(-2), head_size_v = v.size(-1);
  TORCH_CHECK(head_size <= 512, "FlashAttention forward only supports head dimension at most ", 512);
  TORCH_CHECK(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query");
  TORCH_CHECK(head_size % 8 == 0, "head_size should be a multiple of 8");
  TORCH_CHECK(head_size_v == head_size, "fwd: head_size_v must equal head_size on this build");
  TORCH_CHECK(k.size(-1) == head_size, "key must have the head size of query");

  const bool paged = page_table.has_value();
  const bool cache_rows = !paged && k.dim() == 4;  // [slots, seqlen_cache, h_k, d] (reference mha_fwd_nopage)
  const int kv_layout = paged ? 1 : cache_rows ? 2 : 0;
  auto batch_array = [&](const std::optional<Tensor>& t, const char* name) -> const int32_t* {
    if (!t.has_value()) return nullptr;
    TORCH_CHECK(kv_layout != 0, name, " needs a KV cache (a page table or a 4-D cache)");
    TORCH_CHECK(t->is_cuda() && t->is_contiguous(), name, " must be a contiguous GPU tensor");
    TORCH_CHECK(t->scalar_type() == at::kInt, name, " must have dtype int32");
    TORCH_CHECK(t->dim() == 1 && t->size(0) == batch, name, " must have one entry per sequence");
    return t->data_ptr<int32_t>();
  };
  const int32_t* batch_idx_ptr = batch_array(kv_batch_idx, "kv_batch_idx");
  const int32_t* leftpad_ptr = batch_array(leftpad_k, "leftpad_k");
  const int32_t* table_ptr = nullptr;
  int64_t page_size = 0, table_stride = 0, seqlen_k_max = max_seqlen_k;
  int64_t ks0, ks1, ks2 = 0, vs0, vs1, vs2 = 0;
  if (paged) {
    const Tensor& pt = *page_table;
    CHECK_GPU(pt);
    TORCH_CHECK(pt.scalar_type() == at::kInt, "page_table must have dtype torch.int32");
    TORCH_CHECK(pt.dim() == 2 && pt.stride(-1) == 1, "page_table must have contiguous last dimension");
    TORCH_CHECK(batch_idx_ptr != nullptr || pt.size(0) == batch, "batch_size must be equal to batch_size_k");
    TORCH_CHECK(k.dim() == 4 && v.dim() == 4, "paged key/value must be (num_pages, page_size, h_k, d)");
    TORCH_CHECK(v.size(0) == k.size(0) && v.size(1) == k.size(1) && v.size(2) == num_heads_k,
                "key and value cache shapes differ");
    TORCH_CHECK(cu_seqlens_k.size(0) == batch, "with a page table cu_seqlens_k holds the per-sequence lengths [b]");
    page_size = k.size(1);
    TORCH_CHECK((page_size & (page_size - 1)) == 0, "Unsupported page size for attention: ", page_size);
    table_ptr = pt.data_ptr<int32_t>();
    table_stride = pt.stride(0);
    ks0 = k.stride(0); ks1 = k.stride(1); ks2 = k.stride(2);
    vs0 = v.stride(0); vs1 = v.stride(1); vs2 = v.stride(2);
    if (seqlen_k_max <= 0) seqlen_k_max = pt.size(1) * page_size;
  } else if (cache_rows) {
    TORCH_CHECK(v.dim() == 4 && v.size(0) == k.size(0) && v.size(1) == k.size(1) && v.size(2) == num_heads_k,
                "key and value cache shapes differ");
    TORCH_CHECK(batch_idx_ptr != nullptr || k.size(0) >= batch, "the KV cache must have one row per sequence");
    TORCH_CHECK(cu_seqlens_k.size(0) == batch, "with a KV cache cu_seqlens_k holds the per-sequence lengths [b]");
    ks0 = k.stride(0); ks1 = k.stride(1); ks2 = k.stride(2);
    vs0 = v.stride(0); vs1 = v.stride(1); vs2 = v.stride(2);
    if (seqlen_k_max <= 0) seqlen_k_max = k.size(1);
  } else {
    TORCH_CHECK(k.dim() == 3 && v.dim() == 3, "non-paged key/value must be ragged (total_k, h_k, d) or a 4-D cache");
    TORCH_CHECK(cu_seqlens_k.size(0) == batch + 1, "cu_seqlens_k must have b + 1 entries");
    TORCH_CHECK(v.size(0) == k.size(0) && v.size(1) == num_heads_k, "key and value shapes differ");
    ks0 = k.stride(0); ks1 = k.stride(1);
    vs0 = v.stride(0); vs1 = v.stride(1);
    if (seqlen_k_max <= 0) seqlen_k_max = k.size(0);
  }

  Tensor out;
  if (out_.has_value()) {
    out = *out_;
    TORCH_CHECK(out.scalar_type() == q_type, "out dtype must match q dtype");
    TORCH_CHECK(out.dim() == 3 && out.size(0) == total_q && out.size(1) == num_heads && out.size(2) == head_size_v,
                "out shape must be [total_q, num_heads, head_size_v]");
    TORCH_CHECK(out.device() == q.device(), "out must be on the same device as q");
    TORCH_CHECK(out.stride(-1) == 1, "out must have a contiguous last dimension");
  } else {
    out = at::empty({total_q, num_heads, head_size_v}, q.options());
  }
  Tensor lse = at::empty({num_heads, total_q}, q.options().dtype(at::kFloat));
  Tensor out_accum, lse_accum;
  const float* sinks_ptr = nullptr;
  Tensor sinks_f;
  if (sinks.has_value()) {
    CHECK_GPU(*sinks);
    TORCH_CHECK(sinks->numel() == num_heads, "sinks must have one entry per query head");
    sinks_f = sinks->to(at::kFloat).contiguous();
    sinks_ptr = sinks_f.data_ptr<float>();
  }
  if (batch <= 0 || total_q == 0) return {out, lse, out_accum, lse_accum};

  // num_kv_splits (reference flash_attention.cpp:426-470): -1 or 1 = off, 0 = auto, > 1 = as given. The reference splits
  // automatically only at decode; here "auto" also covers the launches that would leave most of the GPU idle otherwise - a
  // chunk of one long sequence (bs 1, 128 queries over 32768 keys: 32 workgroups, 654 us; with 8 splits 115) - as long as the
  // fp32 partial results stay small
  int64_t splits = 1;
  if (num_kv_splits > 1) splits = num_kv_splits;
  else if (num_kv_splits == 0) {
    splits = sglk_attn_auto_splits(batch, num_heads_k, max_seqlen_q * (num_heads / num_heads_k), seqlen_k_max);
    while (splits > 1 && splits * total_q * num_heads * head_size_v * 4 > (int64_t(256) << 20)) --splits;
  }
  float* po = nullptr;
  float* pl = nullptr;
  if (splits > 1) {
    out_accum = at::empty({splits, total_q, num_heads, head_size_v}, q.options().dtype(at::kFloat));
    lse_accum = at::empty({splits, num_heads, total_q}, q.options().dtype(at::kFloat));
    po = out_accum.data_ptr<float>();
    pl = lse_accum.data_ptr<float>();
  }
  const c10::OptionalDeviceGuard guard(q.device());
  SGLK_CALL(sglk_attn_fwd(stream_of(q), out.data_ptr(), lse.data_ptr<float>(), q.data_ptr(), k.data_ptr(), v.data_ptr(),
                          cu_seqlens_q.data_ptr<int32_t>(), cu_seqlens_k.data_ptr<int32_t>(), table_ptr, sinks_ptr, po, pl,
                          batch, total_q, max_seqlen_q, num_heads, num_heads_k, head_size, page_size, q.stride(0),
                          q.stride(1), out.stride(0), out.stride(1), ks0, ks1, ks2, vs0, vs1, vs2, table_stride,
                          (float)softmax_scale, is_causal ? 1 : 0, window_size_left, window_size_right, (float)softcap,
                          splits, dtype_code(q_type, "q"),
                          fp8_kv ? (k.scalar_type() == at::kFloat8_e4m3fn ? SGLK_FP8_E4M3 : SGLK_FP8_E5M2)
                                 : dtype_code(q_type, "q"),
                          k_descale_ptr, v_descale_ptr, kv_layout, batch_idx_ptr, leftpad_ptr));
  return {out, lse, out_accum, lse_accum};
}

// ---- sgl_per_token_group_quant_8bit_v2 (reference src/sycl/per_token_group_quant_8bit_v2.cpp:714-842) --------

void sgl_per_token_group_quant_8bit_v2(Tensor input, Tensor output_q, Tensor output_s, int64_t group_size, double eps,
                                       double min_8bit, double max_8bit, bool scale_ue8m0, bool fuse_silu_and_mul,
                                       const std::optional<Tensor>& masked_m) {
  CHECK_GPU(input);
  CHECK_GPU(output_q);
  CHECK_GPU(output_s);
  CHECK_CONTIGUOUS(input);
  CHECK_CONTIGUOUS(output_q);
  TORCH_CHECK(input.numel() > 0);
  TORCH_CHECK(std::abs(1e-10 - eps) < 1e-13, "sgl_per_token_group_quant_8bit_v2: eps must be 1e-10");
  TORCH_CHECK(group_size > 0 && input.numel() % group_size == 0, "input.numel() must be divisible by group_size");
  const bool masked_layout = masked_m.has_value();
  TORCH_CHECK(output_s.dim() == (masked_layout ? 3 : 2), "output_s must be ", masked_layout ? 3 : 2, "-D");
  TORCH_CHECK(input.dim() == (masked_layout ? 3 : 2), "input must be ", masked_layout ? 3 : 2, "-D");
  const auto in_t = input.scalar_type();
  TORCH_CHECK(in_t == at::kHalf || in_t == at::kBFloat16, "sgl_per_token_group_quant_8bit_v2: input must be Half or BFloat16");
  const auto q_t = output_q.scalar_type();
  TORCH_CHECK(q_t == at::kChar || q_t == at::kFloat8_e4m3fn, "output_q dtype must be Int8 or Float8_e4m3fn");
  if (q_t == at::kFloat8_e4m3fn) {
    TORCH_CHECK(min_8bit == -448.0 && max_8bit == 448.0, "fp8 limits must be +-448");
  } else {
    TORCH_CHECK(min_8bit == -128.0 && max_8bit == 127.0, "int8 limits must be -128 / 127");
  }
  const int64_t hidden = output_q.size(-1);
  const int64_t rows = output_q.size(-2);
  const int64_t experts = masked_layout ? input.size(0) : 1;
  TORCH_CHECK(hidden % group_size == 0, "the hidden dimension must be divisible by group_size");
  TORCH_CHECK(input.size(-1) == hidden * (fuse_silu_and_mul ? 2 : 1) && input.size(-2) == rows,
              "input / output_q shape mismatch");
  const int64_t groups = hidden / group_size;
  const int32_t* mm = nullptr;
  if (masked_layout) {
    CHECK_GPU(*masked_m);
    TORCH_CHECK(masked_m->scalar_type() == at::kInt && masked_m->numel() == experts && masked_m->is_contiguous(),
                "masked_m must be a contiguous int32 [num_experts] tensor");
    mm = masked_m->data_ptr<int32_t>();
  }
  const bool column_major = output_s.stride(-2) < output_s.stride(-1);
  int kind = 0;
  int64_t s_e = masked_layout ? output_s.stride(0) : 0, s_row = output_s.stride(-2), s_col = output_s.stride(-1);
  if (!scale_ue8m0) {
    TORCH_CHECK(output_s.scalar_type() == at::kFloat, "output_s must be float32 unless scale_ue8m0");
    TORCH_CHECK(output_s.size(-2) == rows && output_s.size(-1) == groups, "output_s must be [.., rows, groups]");
  } else if (column_major) {
    TORCH_CHECK(output_s.element_size() == 4, "column-major ue8m0 scales must be packed 4 per 32-bit element");
    kind = 2;
  } else {
    TORCH_CHECK(output_s.element_size() == 1 && output_s.is_contiguous(), "row-major ue8m0 scales must be a contiguous uint8 tensor");
    kind = 1;
  }
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_per_token_group_quant_8bit_v2(stream_of(input), input.data_ptr(), output_q.data_ptr(), output_s.data_ptr(),
                                               mm, experts, rows, hidden, (int)group_size, (float)eps, (float)min_8bit,
                                               (float)max_8bit, dtype_code(in_t, "input"), dtype_code(q_t, "output_q"),
                                               kind, s_e, s_row, s_col, fuse_silu_and_mul ? 1 : 0));
}

// ---- sampling (reference src/sycl/TopKRenormProbs.cpp:300-330, TopPRenormProbs.cpp, TopKTopPSamplingFromProbs.cpp:265-347,
//      MinPSamplingFromProbs.cpp) ---------------------------------------------------------------------------------

void check_probs(const Tensor& probs, const char* op) {
  CHECK_GPU(probs);
  TORCH_CHECK(probs.is_contiguous(), op, ": probs must be contiguous");
  TORCH_CHECK(probs.dim() == 2, "probs must be a 2D tensor [batch_size, vocab_size]");
  TORCH_CHECK(probs.scalar_type() == at::kFloat, "probs must be float32");
}

template <typename T>
const T* optional_row_array(const std::optional<Tensor>& t, at::ScalarType dt, int64_t batch, const char* name) {
  if (!t.has_value()) return nullptr;
  TORCH_CHECK(t->is_cuda() && t->is_contiguous(), name, " must be a contiguous GPU tensor");
  TORCH_CHECK(t->dim() == 1, name, " must be a 1D tensor");
  TORCH_CHECK(t->scalar_type() == dt, name, " must be ", dt);
  TORCH_CHECK(t->size(0) == batch, name, " size must match batch_size");
  return t->data_ptr<T>();
}

// The generator's Philox stream, one 128-bit block per row. Outside a capture: (seed, offset) as host scalars, as the reference reads
// them. While the stream is capturing: pointers to the state torch keeps on the device for the graph plus this launch's offset
// inside it, so that each replay draws fresh numbers (the generator must be registered with the graph - the default one is).
at::PhiloxCudaState philox_state(const std::optional<at::Generator>& gen) {
  auto* impl = at::get_generator_or_default<at::CUDAGeneratorImpl>(gen, at::cuda::detail::getDefaultCUDAGenerator());
  std::lock_guard<std::mutex> lock(impl->mutex_);
  return impl->philox_cuda_state(4);
}

// A decode batch over a long vocabulary: several workgroups per row with a scratch tensor (include/sglk.h, sglk_sampling_ws).
// Returns false when the shape is not that case and the caller takes the one-workgroup-per-row entry.
bool sampling_cluster(int op, const Tensor& probs, void* result, const int64_t* indices, const void* k_arr, int k_is_i64, int64_t k_val,
                      const float* p_arr, float p_val, int64_t batch, const at::PhiloxCudaState* ph) {
  const int64_t ws_bytes = sglk_sampling_workspace_size(batch, probs.size(1));
  if (ws_bytes <= 0) return false;
  Tensor ws = at::empty({ws_bytes}, probs.options().dtype(at::kByte));
  const bool cap = ph != nullptr && ph->captured_;
  SGLK_CALL(sglk_sampling_ws(stream_of(probs), op, result, probs.data_ptr<float>(), indices, k_arr, k_is_i64, k_val, p_arr, p_val, batch,
                             probs.size(1), (ph && !cap) ? ph->seed_.val : 0, ph ? (cap ? ph->offset_intragraph_ : ph->offset_.val) : 0,
                             cap ? ph->seed_.ptr : nullptr, cap ? ph->offset_.ptr : nullptr, ws.data_ptr(), ws_bytes));
  return true;
}

void top_k_renorm_probs(const Tensor& probs, Tensor& renorm_probs, const std::optional<Tensor>& maybe_top_k_arr, int64_t top_k_val) {
  check_probs(probs, "top_k_renorm_probs");
  CHECK_GPU(renorm_probs);
  CHECK_CONTIGUOUS(renorm_probs);
  TORCH_CHECK(probs.sizes() == renorm_probs.sizes(), "Input tensors must have the same shape");
  TORCH_CHECK(probs.scalar_type() == renorm_probs.scalar_type(), "Input tensors must have the same dtype");
  const int64_t* k = optional_row_array<int64_t>(maybe_top_k_arr, at::kLong, probs.size(0), "maybe_top_k_arr");
  if (!k) TORCH_CHECK(top_k_val > 0, "top_k_val must be positive");
  const c10::OptionalDeviceGuard guard(probs.device());
  if (sampling_cluster(0, probs, renorm_probs.data_ptr<float>(), nullptr, k, 1, top_k_val, nullptr, 0.f, probs.size(0), nullptr)) return;
  SGLK_CALL(sglk_top_k_renorm_probs(stream_of(probs), renorm_probs.data_ptr<float>(), probs.data_ptr<float>(), k, top_k_val,
                                    probs.size(0), probs.size(1)));
}

void top_p_renorm_probs(const Tensor& probs, Tensor& renorm_probs, const std::optional<Tensor>& maybe_top_p_arr, double top_p_val) {
  check_probs(probs, "top_p_renorm_probs");
  CHECK_GPU(renorm_probs);
  CHECK_CONTIGUOUS(renorm_probs);
  TORCH_CHECK(probs.sizes() == renorm_probs.sizes(), "Input tensors must have the same shape");
  TORCH_CHECK(probs.scalar_type() == renorm_probs.scalar_type(), "Input tensors must have the same dtype");
  const float* pa = optional_row_array<float>(maybe_top_p_arr, at::kFloat, probs.size(0), "maybe_top_p_arr");
  if (!pa) TORCH_CHECK(top_p_val > 0.0 && top_p_val <= 1.0, "top_p_val must be within (0, 1]");
  const c10::OptionalDeviceGuard guard(probs.device());
  if (sampling_cluster(1, probs, renorm_probs.data_ptr<float>(), nullptr, nullptr, 0, 0, pa, (float)top_p_val, probs.size(0), nullptr)) return;
  SGLK_CALL(sglk_top_p_renorm_probs(stream_of(probs), renorm_probs.data_ptr<float>(), probs.data_ptr<float>(), pa, (float)top_p_val,
                                    probs.size(0), probs.size(1)));
}

void sampling_common(const char* op, const Tensor& probs, Tensor& output, const std::optional<Tensor>& maybe_indices,
                     int64_t& batch, const int64_t*& indices) {
  check_probs(probs, op);
  CHECK_GPU(output);
  CHECK_CONTIGUOUS(output);
  TORCH_CHECK(output.dim() == 1, "output must be a 1D tensor [batch_size]");
  TORCH_CHECK(output.scalar_type() == at::kInt, "output must be int32");
  batch = output.size(0);
  indices = optional_row_array<int64_t>(maybe_indices, at::kLong, batch, "maybe_indices");
  if (!indices) TORCH_CHECK(probs.size(0) == batch, "probs.size(0) must match output.size(0) when maybe_indices is not provided");
}

void top_k_top_p_sampling_from_probs(Tensor probs, Tensor output, std::optional<Tensor> maybe_indices,
                                     std::optional<Tensor> maybe_top_k_arr, int64_t top_k_val, std::optional<Tensor> maybe_top_p_arr,
                                     double top_p_val, bool deterministic, std::optional<at::Generator> gen) {
  (void)deterministic;  // the kernel is reproducible from (seed, offset) either way
  int64_t batch;
  const int64_t* indices;
  sampling_common("top_k_top_p_sampling_from_probs", probs, output, maybe_indices, batch, indices);
  const int32_t* k = optional_row_array<int32_t>(maybe_top_k_arr, at::kInt, batch, "maybe_top_k_arr");
  if (!k) TORCH_CHECK(top_k_val > 0 && top_k_val <= probs.size(1), "top_k_val must be within (0, vocab_size]");
  const float* pa = optional_row_array<float>(maybe_top_p_arr, at::kFloat, batch, "maybe_top_p_arr");
  if (!pa) TORCH_CHECK(top_p_val > 0.0 && top_p_val <= 1.0, "top_p_val must be within (0, 1]");
  const c10::OptionalDeviceGuard guard(probs.device());
  const auto ph = philox_state(gen);
  if (sampling_cluster(2, probs, output.data_ptr<int32_t>(), indices, k, 0, top_k_val, pa, (float)top_p_val, batch, &ph)) return;
  if (ph.captured_) {
    SGLK_CALL(sglk_top_k_top_p_sampling_from_probs_graph(stream_of(probs), output.data_ptr<int32_t>(), probs.data_ptr<float>(), indices,
                                                         k, top_k_val, pa, (float)top_p_val, 1, batch, probs.size(1), ph.seed_.ptr,
                                                         ph.offset_.ptr, ph.offset_intragraph_));
    return;
  }
  SGLK_CALL(sglk_top_k_top_p_sampling_from_probs(stream_of(probs), output.data_ptr<int32_t>(), probs.data_ptr<float>(), indices, k,
                                                 top_k_val, pa, (float)top_p_val, 1, batch, probs.size(1), ph.seed_.val,
                                                 ph.offset_.val));
}

void top_p_sampling_from_probs(Tensor probs, Tensor output, std::optional<Tensor> maybe_indices, std::optional<Tensor> maybe_top_p_arr,
                               double top_p_val, bool deterministic, std::optional<at::Generator> gen) {
  (void)deterministic;
  int64_t batch;
  const int64_t* indices;
  sampling_common("top_p_sampling_from_probs", probs, output, maybe_indices, batch, indices);
  const float* pa = optional_row_array<float>(maybe_top_p_arr, at::kFloat, batch, "maybe_top_p_arr");
  if (!pa) TORCH_CHECK(top_p_val > 0.0 && top_p_val <= 1.0, "top_p_val must be within (0, 1]");
  const c10::OptionalDeviceGuard guard(probs.device());
  const auto ph = philox_state(gen);
  if (sampling_cluster(3, probs, output.data_ptr<int32_t>(), indices, nullptr, 0, 0, pa, (float)top_p_val, batch, &ph)) return;
  if (ph.captured_) {
    SGLK_CALL(sglk_top_k_top_p_sampling_from_probs_graph(stream_of(probs), output.data_ptr<int32_t>(), probs.data_ptr<float>(), indices,
                                                         nullptr, 0, pa, (float)top_p_val, 0, batch, probs.size(1), ph.seed_.ptr,
                                                         ph.offset_.ptr, ph.offset_intragraph_));
    return;
  }
  SGLK_CALL(sglk_top_k_top_p_sampling_from_probs(stream_of(probs), output.data_ptr<int32_t>(), probs.data_ptr<float>(), indices,
                                                 nullptr, 0, pa, (float)top_p_val, 0, batch, probs.size(1), ph.seed_.val,
                                                 ph.offset_.val));
}

void min_p_sampling_from_probs(const Tensor& probs, Tensor& output, const std::optional<Tensor>& maybe_indices,
                               const std::optional<Tensor>& maybe_min_p_arr, double min_p_val, bool deterministic,
                               const std::optional<at::Generator>& gen) {
  (void)deterministic;
  int64_t batch;
  const int64_t* indices;
  sampling_common("min_p_sampling_from_probs", probs, output, maybe_indices, batch, indices);
  const float* pa = optional_row_array<float>(maybe_min_p_arr, at::kFloat, batch, "maybe_min_p_arr");
  const c10::OptionalDeviceGuard guard(probs.device());
  const auto ph = philox_state(gen);
  if (sampling_cluster(4, probs, output.data_ptr<int32_t>(), indices, nullptr, 0, 0, pa, (float)min_p_val, batch, &ph)) return;
  if (ph.captured_) {
    SGLK_CALL(sglk_min_p_sampling_from_probs_graph(stream_of(probs), output.data_ptr<int32_t>(), probs.data_ptr<float>(), indices, pa,
                                                   (float)min_p_val, batch, probs.size(1), ph.seed_.ptr, ph.offset_.ptr,
                                                   ph.offset_intragraph_));
    return;
  }
  SGLK_CALL(sglk_min_p_sampling_from_probs(stream_of(probs), output.data_ptr<int32_t>(), probs.data_ptr<float>(), indices, pa,
                                           (float)min_p_val, batch, probs.size(1), ph.seed_.val, ph.offset_.val));
}

// ---- DeepSeek-style routers (reference src/sycl/TopKSigMoid.cpp, BiasedTopK.cpp:457-520, MoE_fused_gate.cpp:486-600) ----

void topk_sigmoid(Tensor& topk_weights, Tensor& topk_indices, const Tensor& gating_output, bool renormalize,
                  const std::optional<Tensor>& correction_bias, double routed_scaling_factor, int64_t num_fused_shared_experts) {
  CHECK_GPU(topk_weights);
  CHECK_GPU(topk_indices);
  CHECK_GPU(gating_output);
  CHECK_CONTIGUOUS(topk_weights);
  CHECK_CONTIGUOUS(topk_indices);
  CHECK_CONTIGUOUS(gating_output);
  TORCH_CHECK(gating_output.dim() == 2, "gating_output must be 2D [tokens, experts]");
  TORCH_CHECK(topk_weights.scalar_type() == at::kFloat && topk_indices.scalar_type() == at::kInt,
              "topk_weights must be float32 and topk_indices int32");
  TORCH_CHECK(topk_weights.dim() == 2 && topk_weights.sizes() == topk_indices.sizes() && topk_weights.size(0) == gating_output.size(0),
              "topk_weights / topk_indices must be [tokens, topk]");
  const float* bias = nullptr;
  if (correction_bias.has_value()) {
    CHECK_GPU((*correction_bias));
    TORCH_CHECK(correction_bias->scalar_type() == at::kFloat && correction_bias->is_contiguous() &&
                    correction_bias->numel() == gating_output.size(1),
                "correction_bias must be a contiguous float32 tensor with one entry per expert");
    bias = correction_bias->data_ptr<float>();
  }
  const c10::OptionalDeviceGuard guard(gating_output.device());
  SGLK_CALL(sglk_topk_sigmoid(stream_of(gating_output), topk_weights.data_ptr<float>(), topk_indices.data_ptr<int32_t>(),
                              gating_output.data_ptr(), bias, gating_output.size(0), gating_output.size(1), topk_weights.size(1),
                              renormalize ? 1 : 0, (float)routed_scaling_factor, num_fused_shared_experts,
                              dtype_code(gating_output.scalar_type(), "topk_sigmoid")));
}

void biased_topk(const Tensor& input, const Tensor& bias, Tensor& output, Tensor& indices, int64_t topk, int64_t scoring_func,
                 int64_t num_fused_shared_experts, bool renormalize, double routed_scaling_factor,
                 bool apply_routed_scaling_factor_on_output) {
  CHECK_GPU(input);
  CHECK_GPU(bias);
  CHECK_GPU(output);
  CHECK_GPU(indices);
  CHECK_CONTIGUOUS(input);
  CHECK_CONTIGUOUS(bias);
  CHECK_CONTIGUOUS(output);
  CHECK_CONTIGUOUS(indices);
  TORCH_CHECK(input.dim() == 2, "input must be 2D, got ", input.dim(), "D");
  TORCH_CHECK(bias.dim() == 1, "bias must be 1D, got ", bias.dim(), "D");
  TORCH_CHECK(input.size(1) == bias.size(0), "input.size(1) must match bias.size(0)");
  TORCH_CHECK(input.scalar_type() == at::kFloat || input.scalar_type() == at::kHalf || input.scalar_type() == at::kBFloat16,
              "input must be float32, float16, or bfloat16");
  TORCH_CHECK(bias.scalar_type() == at::kFloat, "bias must be float32");
  TORCH_CHECK(topk > num_fused_shared_experts, "topk must be greater than num_fused_shared_experts");
  TORCH_CHECK(scoring_func == 0 || scoring_func == 1, "scoring_func must be 0 (sigmoid) or 1 (sqrtsoftplus)");
  TORCH_CHECK(output.scalar_type() == at::kFloat, "output must be float32");
  TORCH_CHECK(indices.scalar_type() == at::kInt, "indices must be int32");
  TORCH_CHECK(output.dim() == 2 && output.size(0) == input.size(0) && output.size(1) == topk, "output shape mismatch");
  TORCH_CHECK(indices.dim() == 2 && indices.size(0) == input.size(0) && indices.size(1) == topk, "indices shape mismatch");
  const c10::OptionalDeviceGuard guard(input.device());
  SGLK_CALL(sglk_biased_topk(stream_of(input), output.data_ptr<float>(), indices.data_ptr<int32_t>(), input.data_ptr(),
                             bias.data_ptr<float>(), input.size(0), input.size(1), topk, (int)scoring_func,
                             num_fused_shared_experts, renormalize ? 1 : 0, (float)routed_scaling_factor,
                             apply_routed_scaling_factor_on_output ? 1 : 0, dtype_code(input.scalar_type(), "biased_topk")));
}

std::vector<Tensor> moe_fused_gate(const Tensor& input, const std::optional<Tensor>& bias, int64_t num_expert_group,
                                   int64_t topk_group, int64_t topk, int64_t num_fused_shared_experts, int64_t scoring_func,
                                   bool renormalize, double routed_scaling_factor, bool apply_routed_scaling_factor_on_output) {
  CHECK_GPU(input);
  CHECK_CONTIGUOUS(input);
  TORCH_CHECK(input.dim() == 2, "input must be 2D [tokens, experts]");
  const void* bias_ptr = nullptr;
  if (bias.has_value()) {
    CHECK_GPU((*bias));
    TORCH_CHECK(input.dtype() == bias->dtype(), "input and bias should have the same dtype");
    TORCH_CHECK(bias->dim() == 1, "bias must be a 1D tensor when provided");
    TORCH_CHECK(bias->size(0) == input.size(1), "bias size must match the number of experts, but got ", bias->size(0), " vs ",
                input.size(1));
    TORCH_CHECK(bias->is_contiguous(), "bias must be contiguous");
    bias_ptr = bias->data_ptr();
  }
  TORCH_CHECK(scoring_func == 0 || scoring_func == 1, "scoring_func must be 0 (sigmoid) or 1 (softmax), but got ", scoring_func);
  const int64_t num_experts = input.size(1);
  TORCH_CHECK(num_experts % num_expert_group == 0, "num_experts must be divisible by num_expert_group, but got ", num_experts,
              " / ", num_expert_group);
  const c10::OptionalDeviceGuard guard(input.device());
  Tensor output = at::empty({input.size(0), topk}, input.options().dtype(at::kFloat));
  Tensor indices = at::empty({input.size(0), topk}, input.options().dtype(at::kInt));
  SGLK_CALL(sglk_moe_fused_gate(stream_of(input), output.data_ptr<float>(), indices.data_ptr<int32_t>(), input.data_ptr(), bias_ptr,
                                input.size(0), num_experts, num_expert_group, topk_group, topk, num_fused_shared_experts,
                                (int)scoring_func, renormalize ? 1 : 0, (float)routed_scaling_factor,
                                apply_routed_scaling_factor_on_output ? 1 : 0, dtype_code(input.scalar_type(), "moe_fused_gate")));
  return {output, indices};
}

// ---- merge_state / merge_state_v2 (reference src/sycl/merge_states.cpp:303-361) ------------------------------

void merge_state_impl(const Tensor& v_a, const Tensor& s_a, const Tensor& v_b, const Tensor& s_b, Tensor& v_merged,
                      Tensor& s_merged, bool base2, const char* name) {
  for (const Tensor* t : {&v_a, &s_a, &v_b, &s_b, (const Tensor*)&v_merged, (const Tensor*)&s_merged}) {
    TORCH_CHECK(t->is_cuda(), name, ": all tensors must be GPU (cuda/hip) tensors");
    TORCH_CHECK(t->is_contiguous(), name, ": all tensors must be contiguous");
  }
  TORCH_CHECK(v_a.dim() == 3 && v_b.dim() == 3, name, ": v_a / v_b must be 3D [tokens, heads, head_size]");
  TORCH_CHECK(s_a.dim() == 2 && s_b.dim() == 2, name, ": s_a / s_b must be 2D [tokens, heads]");
  TORCH_CHECK(v_a.sizes() == v_b.sizes() && s_a.sizes() == s_b.sizes(), name, ": the two states must have the same shape");
  TORCH_CHECK(v_a.size(0) == s_a.size(0) && v_a.size(1) == s_b.size(1), name, ": v and s disagree on tokens / heads");
  TORCH_CHECK(v_merged.sizes() == v_a.sizes() && s_merged.sizes() == s_a.sizes(), name, ": outputs must have the input shapes");
  TORCH_CHECK(s_a.scalar_type() == at::kFloat && s_b.scalar_type() == at::kFloat && s_merged.scalar_type() == at::kFloat,
              name, ": s tensors must be float32");
  TORCH_CHECK(v_a.scalar_type() == v_merged.scalar_type() && v_b.scalar_type() == v_merged.scalar_type(), name,
              ": v tensors must share one dtype");
  const at::ScalarType dt = v_merged.scalar_type();
  TORCH_CHECK(dt == at::kFloat || dt == at::kHalf || dt == at::kBFloat16, "Unsupported dtype for ", name, ": ", dt);
  const c10::OptionalDeviceGuard guard(v_a.device());
  SGLK_CALL(sglk_merge_state(stream_of(v_a), v_merged.data_ptr(), s_merged.data_ptr<float>(), v_a.data_ptr(),
                             s_a.data_ptr<float>(), v_b.data_ptr(), s_b.data_ptr<float>(), v_merged.size(0),
                             v_merged.size(1), v_merged.size(2), dtype_code(dt, name), base2 ? 1 : 0));
}
void merge_state(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor v_merged, Tensor s_merged) {
  merge_state_impl(v_a, s_a, v_b, s_b, v_merged, s_merged, true, "merge_state");
}
void merge_state_v2(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor v_merged, Tensor s_merged) {
  merge_state_impl(v_a, s_a, v_b, s_b, v_merged, s_merged, false, "merge_state_v2");
}

// ---- store_cache (reference src/sycl/KVCache.cpp:75-160) ------------------------------------------------------

void store_cache(Tensor& k, Tensor& v, Tensor& k_cache, Tensor& v_cache, Tensor& indices) {
  CHECK_GPU(k);
  CHECK_GPU(v);
  CHECK_GPU(k_cache);
  CHECK_GPU(v_cache);
  CHECK_GPU(indices);
  CHECK_CONTIGUOUS(k_cache);
  CHECK_CONTIGUOUS(v_cache);
  CHECK_CONTIGUOUS(indices);
  TORCH_CHECK(k.dim() == 2, "k must be 2D [num_tokens, row_dim]");
  TORCH_CHECK(v.dim() == 2, "v must be 2D [num_tokens, row_dim]");
  TORCH_CHECK(k_cache.dim() == 2, "k_cache must be 2D [cache_size, row_dim]");
  TORCH_CHECK(v_cache.dim() == 2, "v_cache must be 2D [cache_size, row_dim]");
  TORCH_CHECK(indices.dim() == 1, "indices must be 1D [num_tokens]");
  TORCH_CHECK(k.size(0) == 0 || k.stride(1) == 1, "k rows must be contiguous (k.stride(1) == 1)");
  TORCH_CHECK(v.size(0) == 0 || v.stride(1) == 1, "v rows must be contiguous (v.stride(1) == 1)");
  TORCH_CHECK(v.sizes() == k.sizes(), "v shape must match k shape");
  TORCH_CHECK(v_cache.sizes() == k_cache.sizes(), "v_cache shape must match k_cache shape");
  TORCH_CHECK(k.size(1) == k_cache.size(1), "k row_dim must match k_cache row_dim");
  TORCH_CHECK(indices.size(0) == k.size(0), "indices length must match num_tokens");
  TORCH_CHECK(indices.scalar_type() == at::kLong, "indices must be int64");
  TORCH_CHECK(k.dtype() == v.dtype(), "k and v must have the same dtype");
  TORCH_CHECK(k.dtype() == k_cache.dtype(), "k and k_cache must have the same dtype");
  TORCH_CHECK(k.dtype() == v_cache.dtype(), "k and v_cache must have the same dtype");
  if (k.size(0) == 0) return;
  const int64_t esz = k.element_size();
  const c10::OptionalDeviceGuard guard(k.device());
  SGLK_CALL(sglk_store_cache(stream_of(k), k_cache.data_ptr(), v_cache.data_ptr(), k.data_ptr(), v.data_ptr(),
                             indices.data_ptr<int64_t>(), k.size(0), k.size(1) * esz, k.stride(0) * esz, v.stride(0) * esz));
}

// ---- fused_qk_norm_rope / fused_inplace_qknorm_rope (reference src/sycl/FusedQKNormRope.cpp:507-615, :1723-1861) ----

void fused_qk_norm_rope(Tensor& qkv, int64_t num_heads_q, int64_t num_heads_k, int64_t num_heads_v, int64_t head_dim,
                        double eps, Tensor& q_weight, Tensor& k_weight, double base, bool is_neox, Tensor& position_ids,
                        double factor, double low, double high, double attention_factor, int64_t rotary_dim) {
  TORCH_CHECK(qkv.dim() == 2, "QKV tensor must be 2D: [num_tokens, (num_heads_q+num_heads_k+num_heads_v)*head_dim]");
  TORCH_CHECK(position_ids.dim() == 1, "Position IDs must be 1D: [num_tokens]");
  TORCH_CHECK(q_weight.dim() == 1, "Query weights must be 1D: [head_dim]");
  TORCH_CHECK(k_weight.dim() == 1, "Key weights must be 1D: [head_dim]");
  TORCH_CHECK(q_weight.size(0) == head_dim, "Query weights size must match head dimension");
  TORCH_CHECK(k_weight.size(0) == head_dim, "Key weights size must match head dimension");
  CHECK_GPU(qkv);
  CHECK_CONTIGUOUS(qkv);
  CHECK_GPU(position_ids);
  CHECK_CONTIGUOUS(position_ids);
  TORCH_CHECK(position_ids.scalar_type() == at::kInt, "position_ids must have dtype int32 (at::kInt); got ",
              position_ids.scalar_type());
  CHECK_GPU(q_weight);
  CHECK_CONTIGUOUS(q_weight);
  CHECK_GPU(k_weight);
  CHECK_CONTIGUOUS(k_weight);
  TORCH_CHECK(q_weight.scalar_type() == qkv.scalar_type() && k_weight.scalar_type() == qkv.scalar_type(),
              "q_weight / k_weight must have the dtype of qkv");
  const int64_t num_tokens = qkv.size(0);
  TORCH_CHECK(position_ids.size(0) == num_tokens, "Number of tokens in position_ids must match QKV");
  TORCH_CHECK(qkv.size(1) == (num_heads_q + num_heads_k + num_heads_v) * head_dim,
              "QKV tensor size must match total number of heads and head dimension");
  const at::ScalarType dt = qkv.scalar_type();
  TORCH_CHECK(dt == at::kFloat || dt == at::kHalf || dt == at::kBFloat16, "Unsupported dtype for fused_qk_norm_rope: ", dt);
  const c10::OptionalDeviceGuard guard(qkv.device());
  SGLK_CALL(sglk_fused_qknorm_rope_yarn(stream_of(qkv), qkv.data_ptr(), q_weight.data_ptr(), k_weight.data_ptr(),
                                        position_ids.data_ptr<int32_t>(), num_tokens, num_heads_q, num_heads_k, num_heads_v,
                                        head_dim, rotary_dim, (float)eps, (float)base, is_neox ? 1 : 0, (float)factor,
                                        (float)low, (float)high, (float)attention_factor, dtype_code(dt, "fused_qk_norm_rope")));
}

void fused_inplace_qknorm_rope(Tensor& q, Tensor& k, Tensor& q_weight, Tensor& k_weight, Tensor& cos_sin_cache,
                               Tensor& positions, bool is_neox, double eps, int64_t head_dim, int64_t rope_dim) {
  TORCH_CHECK(q.dim() == k.dim(), "q and k must have the same rank, got q:", q.dim(), " k:", k.dim());
  TORCH_CHECK(q.dim() == 3 || q.dim() == 4, "q/k must be 3D or 4D tensors, got q:", q.dim());
  TORCH_CHECK(q.scalar_type() == k.scalar_type(), "q and k must have the same dtype");
  TORCH_CHECK(q_weight.scalar_type() == q.scalar_type(), "q_weight dtype must match q dtype");
  TORCH_CHECK(k_weight.scalar_type() == k.scalar_type(), "k_weight dtype must match k dtype");
  TORCH_CHECK(cos_sin_cache.scalar_type() == at::kFloat, "cos_sin_cache must be float32");
  CHECK_GPU(q);
  TORCH_CHECK(q.stride(-1) == 1, "q must be contiguous in its last dimension (head_dim)");
  CHECK_GPU(k);
  TORCH_CHECK(k.stride(-1) == 1, "k must be contiguous in its last dimension (head_dim)");
  CHECK_GPU(q_weight);
  CHECK_CONTIGUOUS(q_weight);
  CHECK_GPU(k_weight);
  CHECK_CONTIGUOUS(k_weight);
  CHECK_GPU(cos_sin_cache);
  CHECK_CONTIGUOUS(cos_sin_cache);
  CHECK_GPU(positions);
  CHECK_CONTIGUOUS(positions);
  // a 4-D [batch, seq, heads, D] input is addressed as [batch * seq, heads, D]: batch and seq must merge
  auto flat = [](const Tensor& t, const char* name, int64_t& tokens, int64_t& heads, int64_t& ts, int64_t& hs) {
    if (t.dim() == 4) {
      TORCH_CHECK(t.stride(0) == t.size(1) * t.stride(1), name,
                  " batch and sequence dimensions must be mergeable (i.e. contiguous with each other) for 4D input");
      tokens = t.size(0) * t.size(1); heads = t.size(2); ts = t.stride(1); hs = t.stride(2);
    } else {
      tokens = t.size(0); heads = t.size(1); ts = t.stride(0); hs = t.stride(1);
    }
  };
  int64_t tq, hq, q_ts, q_hs, tk, hk, k_ts, k_hs;
  flat(q, "q", tq, hq, q_ts, q_hs);
  flat(k, "k", tk, hk, k_ts, k_hs);
  TORCH_CHECK(tq == tk, "q and k must have the same token count after flattening");
  TORCH_CHECK(q.size(-1) == k.size(-1), "q and k must have the same head_dim");
  const int64_t inferred_head_dim = q.size(-1);
  TORCH_CHECK(q_weight.dim() == 1, "q_weight must be 1D [head_dim]");
  TORCH_CHECK(k_weight.dim() == 1, "k_weight must be 1D [head_dim]");
  TORCH_CHECK(q_weight.size(0) == inferred_head_dim, "q_weight size must match head_dim");
  TORCH_CHECK(k_weight.size(0) == inferred_head_dim, "k_weight size must match head_dim");
  TORCH_CHECK(cos_sin_cache.dim() == 2, "cos_sin_cache must be 2D [max_position, rope_dim]");
  const int64_t inferred_rope_dim = cos_sin_cache.size(1);
  if (head_dim != 0) TORCH_CHECK(head_dim == inferred_head_dim, "head_dim must match q/k hidden size, got ", head_dim, " vs ", inferred_head_dim);
  if (rope_dim != 0) TORCH_CHECK(rope_dim == inferred_rope_dim, "rope_dim must match cos_sin_cache width, got ", rope_dim, " vs ", inferred_rope_dim);
  TORCH_CHECK(inferred_rope_dim % 2 == 0, "rope_dim must be even");
  TORCH_CHECK(inferred_rope_dim <= inferred_head_dim, "rope_dim must be <= head_dim");
  TORCH_CHECK(positions.dim() == 1, "positions must be 1D [num_tokens]");
  TORCH_CHECK(positions.size(0) == tq, "positions size must match flattened q/k tokens");
  TORCH_CHECK(positions.scalar_type() == at::kInt || positions.scalar_type() == at::kLong,
              "Unsupported dtype for fused_inplace_qknorm_rope positions: ", positions.scalar_type());
  const at::ScalarType dt = q.scalar_type();
  TORCH_CHECK(dt == at::kFloat || dt == at::kHalf || dt == at::kBFloat16, "Unsupported dtype for fused_inplace_qknorm_rope: ", dt);
  const c10::OptionalDeviceGuard guard(q.device());
  SGLK_CALL(sglk_fused_qknorm_rope_cache(stream_of(q), q.data_ptr(), k.data_ptr(), q_weight.data_ptr(), k_weight.data_ptr(),
                                         cos_sin_cache.data_ptr<float>(), positions.data_ptr(),
                                         positions.scalar_type() == at::kLong ? 1 : 0, tq, hq, hk, inferred_head_dim,
                                         inferred_rope_dim, q_ts, q_hs, k_ts, k_hs, is_neox ? 1 : 0, (float)eps,
                                         dtype_code(dt, "fused_inplace_qknorm_rope")));
}

// ---- rotary_embedding (reference src/sycl/Rope.cpp:453-471) -------------------------------------------------

std::tuple<Tensor, Tensor> rotary_embedding(Tensor& positions, Tensor& query, Tensor& key, int64_t head_size,
                                            Tensor& cos_sin_cache, bool is_neox) {
  CHECK_GPU(positions);
  CHECK_GPU(query);
  CHECK_GPU(key);
  CHECK_GPU(cos_sin_cache);
  const auto dim = query.dim();
  TORCH_CHECK(dim == 2 || dim == 3,
              " Query/Key must be 2D [num_tokens, num_heads*head_size] or 3D [num_tokens, num_heads, head_size] tensor");
  TORCH_CHECK(key.dim() == dim, "query and key must have the same rank");
  TORCH_CHECK(positions.scalar_type() == at::kLong, "positions must be int64");
  TORCH_CHECK(cos_sin_cache.dim() == 2 && cos_sin_cache.is_contiguous(), "cos_sin_cache must be contiguous [max_pos, rot_dim]");
  TORCH_CHECK(cos_sin_cache.scalar_type() == query.scalar_type() && key.scalar_type() == query.scalar_type(),
              "query, key and cos_sin_cache must share one dtype");
  CHECK_LAST_DIM_CONTIGUOUS(query);
  CHECK_LAST_DIM_CONTIGUOUS(key);
  const int64_t rot_dim = cos_sin_cache.size(1);
  const Tensor pos = positions.reshape({-1}).contiguous();
  const int64_t tokens = pos.size(0);
  const c10::OptionalDeviceGuard guard(query.device());
  if (dim == 2) {
    TORCH_CHECK(head_size > 0 && query.size(1) % head_size == 0 && key.size(1) % head_size == 0,
                "query/key widths must be multiples of head_size");
    TORCH_CHECK(query.size(0) == tokens && key.size(0) == tokens, "positions and query/key disagree on the token count");
    SGLK_CALL(sglk_rotary_embedding(stream_of(query), query.data_ptr(), key.data_ptr(), query.data_ptr(), key.data_ptr(),
                                    pos.data_ptr<int64_t>(), cos_sin_cache.data_ptr(), tokens, query.size(1) / head_size,
                                    key.size(1) / head_size, head_size, rot_dim, query.stride(0), head_size, key.stride(0),
                                    head_size, query.stride(0), head_size, key.stride(0), head_size, is_neox ? 1 : 0,
                                    dtype_code(query.scalar_type(), "query")));
    return {query, key};
  }
  TORCH_CHECK(cos_sin_cache.size(1) == query.size(2), "Rotary dim doesn't match query head_size");
  TORCH_CHECK(cos_sin_cache.size(1) == key.size(2), "Rotary dim doesn't match key head_size");
  TORCH_CHECK(query.size(0) == tokens && key.size(0) == tokens, "positions and query/key disagree on the token count");
  Tensor q_out = at::empty_like(query), k_out = at::empty_like(key);
  q_out = q_out.contiguous();
  k_out = k_out.contiguous();
  SGLK_CALL(sglk_rotary_embedding(stream_of(query), q_out.data_ptr(), k_out.data_ptr(), query.data_ptr(), key.data_ptr(),
                                  pos.data_ptr<int64_t>(), cos_sin_cache.data_ptr(), tokens, query.size(1), key.size(1),
                                  query.size(2), rot_dim, query.stride(0), query.stride(1), key.stride(0), key.stride(1),
                                  q_out.stride(0), q_out.stride(1), k_out.stride(0), k_out.stride(1), is_neox ? 1 : 0,
                                  dtype_code(query.scalar_type(), "query")));
  return {q_out, k_out};
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(sgl_kernel, m) {
  // reference src/torch_extension_sycl.cc:29-51
  m.def("silu_and_mul(Tensor! out, Tensor input) -> ()");
  m.impl("silu_and_mul", c10::kCUDA, &silu_and_mul);
  m.def("gelu_tanh_and_mul(Tensor! out, Tensor input) -> ()");
  m.impl("gelu_tanh_and_mul", c10::kCUDA, &gelu_tanh_and_mul);
  m.def("gelu_and_mul(Tensor! out, Tensor input) -> ()");
  m.impl("gelu_and_mul", c10::kCUDA, &gelu_and_mul);
  m.def("silu_and_mul_clamp(Tensor! out, Tensor input, float swiglu_limit) -> ()");  // reference :32
  m.impl("silu_and_mul_clamp", c10::kCUDA, &silu_and_mul_clamp);
  m.def("swiglu_gpt_oss_sigmoid_alpha(Tensor x, float alpha, float limit) -> Tensor");  // reference :108
  m.impl("swiglu_gpt_oss_sigmoid_alpha", c10::kCUDA, &swiglu_gpt_oss_sigmoid_alpha);
  m.def("rmsnorm(Tensor! output, Tensor input, Tensor weight, float eps) -> ()");
  m.impl("rmsnorm", c10::kCUDA, &rmsnorm);
  m.def("fused_add_rmsnorm(Tensor! input, Tensor! residual, Tensor weight, float eps) -> ()");
  m.impl("fused_add_rmsnorm", c10::kCUDA, &fused_add_rmsnorm);
  m.def("gemma_rmsnorm(Tensor! output, Tensor input, Tensor weight, float eps) -> ()");
  m.impl("gemma_rmsnorm", c10::kCUDA, &gemma_rmsnorm);
  m.def("gemma_fused_add_rmsnorm(Tensor! input, Tensor! residual, Tensor weight, float eps) -> ()");
  m.impl("gemma_fused_add_rmsnorm", c10::kCUDA, &gemma_fused_add_rmsnorm);

  // reference src/torch_extension_sycl.cc:395-398
  m.def(
      "sgl_per_token_group_quant_8bit(Tensor input, Tensor output_q, Tensor output_s, int group_size,"
      " float eps, float fp8_min, float fp8_max, bool scale_ue8m0) -> ()");
  m.impl("sgl_per_token_group_quant_8bit", c10::kCUDA, &sgl_per_token_group_quant_8bit);
  // reference src/torch_extension_sycl.cc:399-402
  m.def(
      "sgl_per_token_group_quant_8bit_v2(Tensor input, Tensor output_q, Tensor output_s, int group_size,"
      " float eps, float fp8_min, float fp8_max, bool scale_ue8m0, bool fuse_silu_and_mul, Tensor? masked_m) -> ()");
  m.impl("sgl_per_token_group_quant_8bit_v2", c10::kCUDA, &sgl_per_token_group_quant_8bit_v2);
  // reference src/torch_extension_sycl.cc:117-120
  m.def(
      "rotary_embedding(Tensor positions, Tensor query, Tensor key, int head_size, Tensor cos_sin_cache, "
      "bool is_neox) -> (Tensor, Tensor)");
  m.impl("rotary_embedding", c10::kCUDA, &rotary_embedding);

  // reference src/torch_extension_sycl.cc:122-125
  m.def(
      "store_cache(Tensor k, Tensor v, Tensor(a!) k_cache, Tensor(b!) v_cache, "
      "Tensor indices) -> ()");
  m.impl("store_cache", c10::kCUDA, &store_cache);
  // reference src/torch_extension_sycl.cc:66-80 (top_p_sampling_from_probs: declared at include/sgl_kernel_ops.h:938-945 and
  // called by python/sgl_kernel/sampling.py:119, never registered there; schema authored from the declaration)
  m.def("top_k_renorm_probs(Tensor probs, Tensor! renorm_probs, Tensor? maybe_top_k_arr, int top_k_val) -> ()");
  m.impl("top_k_renorm_probs", c10::kCUDA, &top_k_renorm_probs);
  m.def("top_p_renorm_probs(Tensor probs, Tensor! renorm_probs, Tensor? maybe_top_p_arr, float top_p_val) -> ()");
  m.impl("top_p_renorm_probs", c10::kCUDA, &top_p_renorm_probs);
  m.def(
      "top_k_top_p_sampling_from_probs(Tensor probs, Tensor! output, Tensor? maybe_indices, Tensor? "
      "maybe_top_k_arr, int top_k_val, Tensor? maybe_top_p_arr, float top_p_val, bool deterministic, Generator? "
      "gen) -> ()");
  m.impl("top_k_top_p_sampling_from_probs", c10::kCUDA, &top_k_top_p_sampling_from_probs);
  m.def(
      "top_p_sampling_from_probs(Tensor probs, Tensor! output, Tensor? maybe_indices, Tensor? "
      "maybe_top_p_arr, float top_p_val, bool deterministic, Generator? gen) -> ()");
  m.impl("top_p_sampling_from_probs", c10::kCUDA, &top_p_sampling_from_probs);
  m.def(
      "min_p_sampling_from_probs(Tensor probs, Tensor! output, Tensor? maybe_indices, Tensor? "
      "maybe_min_p_arr, float min_p_val, bool deterministic, Generator? gen) -> ()");
  m.impl("min_p_sampling_from_probs", c10::kCUDA, &min_p_sampling_from_probs);
  // reference src/torch_extension_sycl.cc:55-58, :111-115, :191-196
  m.def(
      "topk_sigmoid(Tensor! topk_weights, Tensor! topk_indices, Tensor gating_output, bool renormalize, Tensor? "
      "correction_bias, float routed_scaling_factor=1.0, int num_fused_shared_experts=0) -> ()");
  m.impl("topk_sigmoid", c10::kCUDA, &topk_sigmoid);
  m.def(
      "biased_topk(Tensor input, Tensor bias, Tensor! output, Tensor! indices, int topk, int scoring_func, int "
      "num_fused_shared_experts, bool renormalize, float routed_scaling_factor, bool "
      "apply_routed_scaling_factor_on_output) -> ()");
  m.impl("biased_topk", c10::kCUDA, &biased_topk);
  m.def(
      "moe_fused_gate(Tensor input, Tensor? bias, int num_expert_group, int topk_group, int topk, int "
      "num_fused_shared_experts, int scoring_func, bool renormalize, float routed_scaling_factor, bool "
      "apply_routed_scaling_factor_on_output) -> "
      "(Tensor[])");
  m.impl("moe_fused_gate", c10::kCUDA, &moe_fused_gate);
  // reference src/torch_extension_sycl.cc:232-235
  m.def("merge_state_v2(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()");
  m.impl("merge_state_v2", c10::kCUDA, &merge_state_v2);
  m.def("merge_state(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()");
  m.impl("merge_state", c10::kCUDA, &merge_state);
  // reference src/torch_extension_sycl.cc:416-424
  m.def(
      "fused_qk_norm_rope(Tensor! qkv, int num_heads_q, int num_heads_k, int num_heads_v, int head_dim, "
      "float eps, Tensor! q_weight, Tensor! k_weight, float base, bool is_neox, Tensor! position_ids, "
      "float factor, float low, float high, float attention_factor, int rotary_dim) -> ()");
  m.impl("fused_qk_norm_rope", c10::kCUDA, &fused_qk_norm_rope);
  m.def(
      "fused_inplace_qknorm_rope(Tensor! q, Tensor! k, Tensor q_weight, Tensor k_weight, "
      "Tensor cos_sin_cache, Tensor positions, bool is_neox, float eps, int head_dim=0, int rope_dim=0) -> ()");
  m.impl("fused_inplace_qknorm_rope", c10::kCUDA, &fused_inplace_qknorm_rope);

  // authored: reference include/sgl_kernel_ops.h:581-586 + python/sgl_kernel/gemm.py:24-31
  m.def(
      "fp8_blockwise_scaled_mm(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype)"
      " -> Tensor");
  m.impl("fp8_blockwise_scaled_mm", c10::kCUDA, &fp8_blockwise_scaled_mm);
  // authored: reference include/sgl_kernel_ops.h:567-580 + python/sgl_kernel/gemm.py:13-42
  m.def(
      "fp8_scaled_mm(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype,"
      " Tensor? bias) -> Tensor");
  m.impl("fp8_scaled_mm", c10::kCUDA, &fp8_scaled_mm);
  m.def(
      "int8_scaled_mm(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype,"
      " Tensor? bias) -> Tensor");
  m.impl("int8_scaled_mm", c10::kCUDA, &int8_scaled_mm);

  // reference src/torch_extension_sycl.cc:328-358
  m.def(
      "fwd(Tensor   q,"
      "    Tensor   k,"
      "    Tensor   v,"
      "    Tensor?  q_v,"
      "    Tensor  cu_seqlens_q,"
      "    Tensor  cu_seqlens_k,"
      "    int     max_seqlen_q,"
      "    int     max_seqlen_k,"
      "    Tensor?  page_table,"
      "    Tensor?  kv_batch_idx,"
      "    Tensor?  leftpad_k,"
      "    Tensor?  rotary_cos,"
      "    Tensor?  rotary_sin,"
      "    Tensor?  seqlens_rotary,"
      "    Tensor?  q_descale,"
      "    Tensor?  k_descale,"
      "    Tensor?  v_descale,"
      "    float    softmax_scale,"
      "    Tensor?  sinks,"
      "    bool     is_causal,"
      "    int      window_size_left,"
      "    int      window_size_right,"
      "    float    softcap,"
      "    bool     is_rotary_interleaved,"
      "    Tensor?  scheduler_metadata,"
      "    int      num_kv_splits,"
      "    bool?    pack_gqa,"
      "    int      sm_margin,"
      "    Tensor(a!)?  out=None) -> (Tensor(a!), Tensor, Tensor, Tensor)");
  m.impl("fwd", c10::kCUDA, &mha_fwd);

  // reference src/torch_extension_sycl.cc:362-368
  m.def("awq_dequantize(Tensor qweight, Tensor scales, Tensor qzeros) -> Tensor");
  m.impl("awq_dequantize", c10::kCUDA, &awq_dequantize);
  m.def("sgl_per_tensor_quant_fp8(Tensor input, Tensor output_q, Tensor output_s, bool is_static) -> ()");
  m.impl("sgl_per_tensor_quant_fp8", c10::kCUDA, &sgl_per_tensor_quant_fp8);
  m.def("sgl_per_token_quant_fp8(Tensor input, Tensor(a!) output_q, Tensor(b!) output_s) -> ()");
  m.impl("sgl_per_token_quant_fp8", c10::kCUDA, &sgl_per_token_quant_fp8);
  m.def(
      "qserve_w4a8_per_chn_gemm(Tensor _in_feats, Tensor _kernel, Tensor _wscales, Tensor _ascales, Tensor _w_szs, "
      "Tensor _a_ssums, Tensor! _out_feats) -> ()");
  m.impl("qserve_w4a8_per_chn_gemm", c10::kCUDA, &qserve_w4a8_per_chn_gemm);
  m.def(
      "qserve_w4a8_per_group_gemm(Tensor _in_feats, Tensor _kernel, Tensor _zeros, Tensor _scales_i8, Tensor _wscales, "
      "Tensor _ascales, Tensor! _out_feats) -> ()");
  m.impl("qserve_w4a8_per_group_gemm", c10::kCUDA, &qserve_w4a8_per_group_gemm);
  m.def("flash_mla_get_workspace_size", &flash_mla_get_workspace_size);
  m.def(
      "flash_mla_decode(Tensor! out, Tensor! q_nope, Tensor! q_pe, Tensor! kv_c_and_k_pe_cache, Tensor! seq_lens,"
      " Tensor! page_table, Tensor! workspace, float sm_scale, int num_kv_splits) -> ()");
  m.impl("flash_mla_decode", c10::kCUDA, &flash_mla_decode);
  m.def("flash_mla_prefill_get_workspace_size", &flash_mla_prefill_get_workspace_size);
  m.def(
      "flash_mla_prefill(Tensor! out, Tensor! q_nope, Tensor! q_pe, Tensor! kv_c_and_k_pe_cache, "
      "Tensor! cu_seqlens_q, Tensor! seq_lens, int max_seqlen_q, "
      "Tensor! page_table, Tensor! workspace, float sm_scale, bool causal, int num_kv_splits) -> ()");
  m.impl("flash_mla_prefill", c10::kCUDA, &flash_mla_prefill);

  // reference src/torch_extension_sycl.cc:53, :199-203, :214-229
  m.def("topk_softmax(Tensor! topk_weights, Tensor! topk_indices, Tensor gating_output, bool renormalize) -> ()");
  m.impl("topk_softmax", c10::kCUDA, &topk_softmax);
  m.def(
      "moe_align_block_size(Tensor topk_ids, int num_experts, int block_size, Tensor! sorted_token_ids, Tensor! "
      "experts_ids, Tensor! num_tokens_post_pad, Tensor! cumsum_buffer, bool "
      "pad_sorted_token_ids) -> ()");
  m.impl("moe_align_block_size", c10::kCUDA, &moe_align_block_size);
  m.def(
      "moe_grouped_mm_nt_xe20(Tensor! output, Tensor activations, Tensor weights, Tensor? bias, Tensor "
      "total_rows_for_experts, int n_experts, int activation_type, bool fuse_act, float gemm1_alpha=1.702, float "
      "gemm1_limit=7.0) -> ()");
  m.impl("moe_grouped_mm_nt_xe20", c10::kCUDA, &moe_grouped_mm_nt_xe20);
  m.def(
      "moe_grouped_mm_nt_xe20_w4a16(Tensor! output, Tensor activations, Tensor packed_weights, Tensor scales, "
      "Tensor? zeros, Tensor? bias, Tensor rows_per_expert, int n_experts, bool is_int4, int group_size) -> ()");
  m.impl("moe_grouped_mm_nt_xe20_w4a16", c10::kCUDA, &moe_grouped_mm_nt_xe20_w4a16);
  m.def(
      "moe_grouped_mm_nt_w4a16_act(Tensor! output, Tensor activations, Tensor packed_weights, Tensor scales, "
      "Tensor? zeros, Tensor? bias, Tensor rows_per_expert, int n_experts, bool is_int4, int group_size, "
      "int activation_type, float act_limit=0.0, Tensor? row_map=None, float act_alpha=0.0) -> ()");
  m.impl("moe_grouped_mm_nt_w4a16_act", c10::kCUDA, &moe_grouped_mm_nt_w4a16_act);
  m.def(
      "moe_grouped_mm_nt_w4a16_splitk(Tensor! output, Tensor! ws, Tensor activations, Tensor packed_weights, Tensor scales, "
      "Tensor? zeros, Tensor rows_per_expert, int n_experts, bool is_int4, int group_size) -> int");
  m.impl("moe_grouped_mm_nt_w4a16_splitk", c10::kCUDA, &moe_grouped_mm_nt_w4a16_splitk);
  m.def("moe_w4a16_splitk_applies(int total_m, int n_experts, int n, int k, int group_size, bool is_int4, bool is_bf16) -> int",
        &moe_w4a16_splitk_applies);
  m.def(
      "apply_shuffle_mul_sum_splitk(Tensor y, Tensor ws, Tensor! output, Tensor permutation, Tensor rows_per_expert, int "
      "block_rows, float routed_scaling_factor, Tensor? factors) -> ()");
  m.impl("apply_shuffle_mul_sum_splitk", c10::kCUDA, &apply_shuffle_mul_sum_splitk);
  m.def(
      "prepare_moe_input(Tensor topk_ids, Tensor! expert_offsets, Tensor? blockscale_offsets, Tensor! problem_sizes1,"
      " Tensor! problem_sizes2, Tensor! input_permutation, Tensor! output_permutation, int num_experts, int n, int k)"
      " -> ()");
  m.impl("prepare_moe_input", c10::kCUDA, &prepare_moe_input);
  m.def("scatter_tokens_to_experts(Tensor input, Tensor src2dst_map, Tensor! output) -> ()");
  m.impl("scatter_tokens_to_experts", c10::kCUDA, &scatter_tokens_to_experts);
  m.def(
      "apply_shuffle_mul_sum(Tensor input, Tensor! output, Tensor permutation, float routed_scaling_factor, Tensor? "
      "factors) -> ()");
  m.impl("apply_shuffle_mul_sum", c10::kCUDA, &apply_shuffle_mul_sum);
}

// The loader does `from sgl_kernel import common_ops` (reference python/sgl_kernel/__init__.py:14);
// registration above runs at dlopen, the module itself is empty (reference sgl_kernel_ops.h:37-41).
extern "C" __attribute__((visibility("default"))) PyObject* PyInit_common_ops(void) {
  static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "common_ops", nullptr, 0, nullptr};
  return PyModule_Create(&module);
}
