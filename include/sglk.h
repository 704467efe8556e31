/*
 * sglk.h — C-ABI of the MI355X (gfx950) sgl_kernel hot-path kernels.
 *
 * This is the drop-in boundary below the torch operator registry: one plain-C
 * entry point per device kernel family, taking a HIP stream, raw device
 * pointers and sizes. No torch types cross this line. The reference's
 * precedent for this shape of interface is its JIT C API
 * (reference include/sgl_kernel/jit_kernel/elementwise/activation.hpp:183-187:
 * `extern "C" void act_and_mul_forward_<dtype>(void* queue, const void* in,
 * void* out, int64 tokens, int64 dim, int32 act_kind)`).
 *
 * Each entry point cites the reference interface it replaces (file:line,
 * relative to the reference checkout). The torch-level schemas that sit on
 * top of these live in sgl-kernel-xpu_amd/csrc/torch_extension_hip.cc and
 * mirror reference src/torch_extension_sycl.cc.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SGLK_E* code otherwise
 *     and never throws; sglk_last_error() returns a thread-local message.
 *   - all pointers are DEVICE pointers unless the name ends in _host.
 *   - work is enqueued on `stream` (a hipStream_t) and is not waited for; no
 *     function allocates, frees or synchronises (graph-capture safe).
 *   - strides are in ELEMENTS unless the name says _bytes.
 */
#ifndef SGLK_H_
#define SGLK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGLK_API __attribute__((visibility("default")))

typedef void* sglk_stream_t; /* hipStream_t */

enum sglk_dtype {
  SGLK_F32 = 0,
  SGLK_F16 = 1,
  SGLK_BF16 = 2,
  SGLK_FP8_E4M3 = 3, /* OCP e4m3fn */
  SGLK_INT8 = 4,
  SGLK_FP8_E5M2 = 5,
  SGLK_U8 = 6,
  SGLK_I32 = 7,
  SGLK_I64 = 8,
};

enum sglk_status {
  SGLK_OK = 0,
  SGLK_EINVAL = -1,       /* bad argument (the torch layer turns this into RuntimeError) */
  SGLK_EUNSUPPORTED = -2, /* shape/dtype combination has no kernel */
  SGLK_ELAUNCH = -3,      /* hipLaunch failed */
};

/* Human-readable reason for the last non-zero return on this thread. */
SGLK_API const char* sglk_last_error(void);
/* Library version string, and the gfx arch the code objects were built for. */
SGLK_API const char* sglk_version(void);
SGLK_API const char* sglk_arch(void);
/* ABI revision of this header. It is raised whenever an entry point changes its parameter list or a workspace its layout or
 * size (5: sglk_moe_grouped_mm_w4a16_act takes row_map / src_rows; the flash_mla_decode workspace holds 16-byte merge
 * counters and, above 64 heads, 128-row fragment slabs). A caller built against another revision must not call in: compare
 * sglk_abi_version() with the SGLK_ABI_VERSION it was compiled with. */
#define SGLK_ABI_VERSION 5
SGLK_API int sglk_abi_version(void);

/* Row addressing of a 2-D / 3-D last-dim-contiguous tensor:
 * offset(row) = (row / inner_size) * outer_stride + (row % inner_size) * inner_stride.
 * Mirrors RowStrides of reference src/sycl/RMSNorm.cpp:44-64. */
typedef struct sglk_row_strides {
  int64_t outer_stride;
  int64_t inner_size;
  int64_t inner_stride;
} sglk_row_strides;

/* ---- RMSNorm family -------------------------------------------------------
 * rmsnorm / gemma_rmsnorm: reference src/sycl/RMSNorm.cpp:793-825, :850-878
 * (schemas src/torch_extension_sycl.cc:41,47).
 *   out[r,:] = T((w * rsqrt(mean(x[r,:]^2) + eps)) * x[r,:])   (gemma: (1+w))
 * x/out dtype in {F32,F16,BF16}; weight dtype in {F32,F16,BF16}. */
SGLK_API int sglk_rmsnorm(sglk_stream_t stream, void* out, const void* x, const void* weight,
                          int64_t rows, int64_t n, sglk_row_strides x_strides,
                          sglk_row_strides out_strides, float eps, int dtype, int weight_dtype,
                          int gemma);

/* fused_add_rmsnorm / gemma_fused_add_rmsnorm: reference
 * src/sycl/RMSNorm.cpp:827-848, :880-905 (schemas torch_extension_sycl.cc:44,50).
 *   r = T(x + residual); residual = r; x = T((w * rstd(r)) * r), contiguous [rows,n]. */
SGLK_API int sglk_fused_add_rmsnorm(sglk_stream_t stream, void* x, void* residual,
                                    const void* weight, int64_t rows, int64_t n, float eps,
                                    int dtype, int weight_dtype, int gemma);

/* ---- activation-and-mul ---------------------------------------------------
 * silu_and_mul / gelu_tanh_and_mul / gelu_and_mul: reference
 * src/sycl/TripleOps.cpp:140-153, :181, :222 (schemas :29,:35,:38).
 *   out[t, j] = T(act(x[t, j]) * x[t, d + j]),  x is [tokens, 2d] contiguous. */
enum sglk_act { SGLK_ACT_SILU = 0, SGLK_ACT_GELU_TANH = 1, SGLK_ACT_GELU = 2 };
SGLK_API int sglk_act_and_mul(sglk_stream_t stream, void* out, const void* x, int64_t tokens,
                              int64_t d, int dtype, int act);

/* silu_and_mul_clamp (the DeepSeek-V4 swiglu): reference src/sycl/SiluAndMulClamp.cpp:55-180 (schema
 * torch_extension_sycl.cc:32), python/sgl_kernel/elementwise.py:231-255. x is [tokens, 2d] contiguous, gate half first:
 *   g = bf16(min(bf16(x[t, j]), bf16(limit)));  u = bf16(clamp(bf16(x[t, d + j]), -bf16(limit), bf16(limit)));
 *   out[t, j] = T(g * sigmoid(g) * u)           (the clamp is done in bf16 whatever T is; T is f16 or bf16) */
SGLK_API int sglk_silu_and_mul_clamp(sglk_stream_t stream, void* out, const void* x, int64_t tokens, int64_t d,
                                     int dtype, float limit);

/* swiglu_gpt_oss_sigmoid_alpha: reference src/sycl/SwigluAlphaLimit.cpp:16-175 (schema torch_extension_sycl.cc:108),
 * called by fused_experts for gemm1_alpha (python/sgl_kernel/moe.py:692-697, :787-789). x is [rows, 2 hidden]
 * contiguous with gate / up INTERLEAVED (x[r, 2 j] = gate, x[r, 2 j + 1] = up); fp32 arithmetic, one rounding to T:
 *   g = min(gate, limit);  u = clamp(up, -limit, limit);  out[r, j] = T(g * sigmoid(alpha * g) * (u + 1)) */
SGLK_API int sglk_swiglu_alpha_limit(sglk_stream_t stream, void* out, const void* x, int64_t rows, int64_t hidden,
                                     int dtype, float alpha, float limit);

/* ---- per-token-group 8-bit quantisation -----------------------------------
 * sgl_per_token_group_quant_8bit: reference
 * src/sycl/per_token_group_quant_8bit.cpp:222-386 (schema :395-398).
 *   per (row, group): amax = max(|x|, eps); s = amax / qmax;
 *   [ue8m0: s = 2^ceil(log2(max(s,1e-10)))]; q = cast(clamp(x * (1/s), qmin, qmax)).
 * x [rows, k] contiguous, F32/F16/BF16. q [rows,k] FP8_E4M3 (RN-even) or INT8
 * (truncating). Scales:
 *   scale_kind 0: float, element (row, g) at  row*s_stride_row + g*s_stride_col
 *   scale_kind 1: ue8m0 bytes, row-major contiguous [rows, k/group]
 *   scale_kind 2: ue8m0 packed 4-per-int32, column-major: byte address
 *                 ((g/4)*s_stride_col + row)*4 + g%4   (s_stride_col in int32 units)
 */
SGLK_API int sglk_per_token_group_quant_8bit(sglk_stream_t stream, const void* x, void* q,
                                             void* scales, int64_t rows, int64_t k,
                                             int group_size, float eps, float qmin, float qmax,
                                             int in_dtype, int out_dtype, int scale_kind,
                                             int64_t s_stride_row, int64_t s_stride_col);

/* ---- fp8 block-scaled GEMM --------------------------------------------------
 * fp8_blockwise_scaled_mm: declared (never implemented) in reference
 * include/sgl_kernel_ops.h:581-586; wrapper python/sgl_kernel/gemm.py:24-31;
 * semantics pinned by tests/test_fp8_blockwise_gemm.py:23-85.
 *   out[m,n] = T( sum_kb sa[m,kb] * sb[kb, n/128] * sum_{k in kb} a[m,k]*b[k,n] )
 * a [M,K] e4m3 row-major (lda); b given as [N,K] K-contiguous (ldb) — i.e. the
 * reference's column-major [K,N]; sa [M, K/128] fp32 with element strides
 * (sa_stride_m, sa_stride_k); sb [K/128, N/128] fp32 (sb_stride_k, sb_stride_n).
 * out [M,N] row-major (ldc), BF16 or F16. K % 128 == 0, N % 16 == 0. */
SGLK_API int sglk_fp8_blockwise_scaled_mm(sglk_stream_t stream, void* out, const void* a,
                                          const void* b, const float* sa, const float* sb,
                                          int64_t M, int64_t N, int64_t K, int64_t lda,
                                          int64_t ldb, int64_t ldc, int64_t sa_stride_m,
                                          int64_t sa_stride_k, int64_t sb_stride_k,
                                          int64_t sb_stride_n, int out_dtype);
/* The same product with a scratch buffer the caller owns (the reference's op allocates nothing and has no counterpart):
 * few rows over a deep K (M = 65 .. 512 with at most 256 / S half tiles of 128 x 256 - Llama-3-8B's down projection at
 * N = 4096, K = 14336) run as tile x K-slice units that store fp32 partial tiles into S slabs of [M, N], added in
 * slice order and rounded once by a second kernel. _workspace_size returns the bytes the shape can use (0: the plain
 * entry's path is taken whatever is passed); a NULL or smaller workspace is not an error - the call then runs unsplit.
 * Results are a pure function of the inputs for a given workspace size class (deterministic, graph-capturable). */
SGLK_API int64_t sglk_fp8_blockwise_scaled_mm_workspace_size(int64_t M, int64_t N, int64_t K);
SGLK_API int sglk_fp8_blockwise_scaled_mm_ws(sglk_stream_t stream, void* out, const void* a, const void* b,
                                             const float* sa, const float* sb, int64_t M, int64_t N, int64_t K,
                                             int64_t lda, int64_t ldb, int64_t ldc, int64_t sa_stride_m,
                                             int64_t sa_stride_k, int64_t sb_stride_k, int64_t sb_stride_n,
                                             int out_dtype, void* workspace, int64_t workspace_bytes);

/* ---- per-token / per-channel scaled GEMM ------------------------------------
 * fp8_scaled_mm / int8_scaled_mm: declared in reference
 * include/sgl_kernel_ops.h:567-580; wrappers python/sgl_kernel/gemm.py:13-42;
 * semantics pinned by tests/test_fp8_gemm.py:11-19 and tests/test_int8_gemm.py:16-22.
 *   fp8 : out = T(T(acc * sa[m] * sb[n]) + bias[n])      (bias added after the cast)
 *   int8: out = T(float(acc_i32) * sa[m] * sb[n] + bias[n])  (bias added in fp32)
 * a [M,K] row-major; b as [N,K] K-contiguous; K % 16 == 0. */
SGLK_API int sglk_scaled_mm(sglk_stream_t stream, void* out, const void* a, const void* b,
                            const float* sa, const float* sb, const void* bias, int64_t M,
                            int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                            int in_dtype, int out_dtype);
/* The same products with a scratch buffer the caller owns (as sglk_fp8_blockwise_scaled_mm_ws above): 129 .. 1024 rows over a
 * deep K run as tile x K-slice units that store their raw accumulators (fp32; int32 for int8 inputs) into slabs of
 * [M, N]; a second kernel adds the slabs and applies row scale, column scale and bias in the order stated above - int8
 * results are bit-identical to the unsplit path (integer sums), fp8 results differ by the fp32 association of the K sum.
 * _workspace_size: bytes the shape can use (0: unsplit whatever is passed). */
SGLK_API int64_t sglk_scaled_mm_workspace_size(int64_t M, int64_t N, int64_t K);
SGLK_API int sglk_scaled_mm_ws(sglk_stream_t stream, void* out, const void* a, const void* b, const float* sa,
                               const float* sb, const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda,
                               int64_t ldb, int64_t ldc, int in_dtype, int out_dtype, void* workspace,
                               int64_t workspace_bytes);

/* ---- quantisation steps around the scaled GEMMs ---------------------------------
 * sgl_per_token_quant_fp8: reference src/sycl/per_token_quant_fp8.cpp:201 (schema torch_extension_sycl.cc:410).
 *   input [rows, cols] F16/BF16/F32 contiguous -> output_q e4m3fn [rows, cols], output_s fp32 [rows]
 *   scale = rowmax|x| / 448, q = e4m3(clamp(x / scale, +-448)) (all zeros for an all-zero row).
 * sgl_per_tensor_quant_fp8: reference src/sycl/per_tensor_quant_fp8.cpp:161 (schema :407).
 *   is_static == 0: output_s[0] (zero-initialised by the caller) is raised to max|x| / 448 first;
 *   q = e4m3(clamp(x * 1 / (output_s[0] + 1e-8), +-448)).
 * awq_dequantize: reference src/sycl/awq_dequantize.cpp:98-123 (schema :26).
 *   qweight int32 [K, C], qzeros int32 [K / group, C], scales F16/BF16 [K / group, 8 C] -> out [K, 8 C] (dtype of scales),
 *   nibble order 0,4,1,5,2,6,3,7 inside each int32. */
SGLK_API int sglk_per_token_quant_fp8(sglk_stream_t stream, void* output_q, float* output_s, const void* input,
                                      int64_t rows, int64_t cols, int dtype);
SGLK_API int sglk_per_tensor_quant_fp8(sglk_stream_t stream, void* output_q, float* output_s, const void* input,
                                       int64_t numel, int is_static, int dtype);
SGLK_API int sglk_awq_dequantize(sglk_stream_t stream, void* out, const int32_t* qweight, const void* scales,
                                 const int32_t* qzeros, int64_t K, int64_t C, int64_t group_size, int dtype);

/* ---- QServe W4A8 GEMMs ------------------------------------------------------
 * Declared only in the reference (include/sgl_kernel_ops.h:1132-1148; wrappers python/sgl_kernel/gemm.py:314-356);
 * meaning, quantisers and the 32x32 interleaved weight packing pinned by tests/test_qserve_w4a8_per_chn_gemm.py:
 * 12-56,80-88 and tests/test_qserve_w4a8_per_group_gemm.py:12-92,134-145.
 *   in_feats [M,K] int8 (row stride lda); kernel [N, K/2] int8 packed; out [M,N] fp16 (row stride ldc)
 *   per_chn  : out = (in @ Wq^T) * ascales[m] * wscales[n] - a_ssums[m] * w_szs[n]       (all scale vectors fp16)
 *   per_group: out = (in @ W8^T) * ascales[m] * wscales[n],  W8 = Wq * scales_i8[k/128, n'] + zeros[k/128, n']
 *              (scales_i8 / zeros int8 [K/128, N] in the permuted column order of the reference packer)
 * N % 32 == 0; K % 64 == 0 (per_chn) or K % 128 == 0 (per_group). */
SGLK_API int sglk_qserve_w4a8_per_chn_gemm(sglk_stream_t stream, void* out, const void* in_feats,
                                           const void* kernel, const void* wscales, const void* ascales,
                                           const void* w_szs, const void* a_ssums, int64_t M, int64_t N,
                                           int64_t K, int64_t lda, int64_t ldc);
SGLK_API int sglk_qserve_w4a8_per_group_gemm(sglk_stream_t stream, void* out, const void* in_feats,
                                             const void* kernel, const void* zeros, const void* scales_i8,
                                             const void* wscales, const void* ascales, int64_t M, int64_t N,
                                             int64_t K, int64_t lda, int64_t ldc);

/* ---- MLA decode -------------------------------------------------------------
 * flash_mla_decode: reference src/sycl/mla_decode.cpp:135-175 (schema
 * src/torch_extension_sycl.cc:364-368; wrapper python/sgl_kernel/attention.py:54-132).
 *   out[b,h,:512] = softmax(sm_scale * [q_nope[b,h], q_pe[b,h]] . C_b^T) . C_b[:, :512]
 * where C_b is the first seq_lens[b] rows of the pages page_table[b,:] of
 * cache [pages, page_size, 576] (16-bit, same dtype as q). q strides in elements;
 * cache_page_stride = elements between consecutive pages (rows are 576 contiguous).
 * num_kv_splits < 1 selects sglk_mla_decode_auto_splits(batch, pages_per_seq*page_size);
 * splits > 1 need `workspace` of sglk_mla_decode_workspace_size(...) bytes.
 * flash_mla_get_workspace_size: reference src/sycl/mla_decode.cpp:192-223. */
SGLK_API int64_t sglk_mla_decode_auto_splits(int64_t batch, int64_t max_seq_len);
SGLK_API int64_t sglk_mla_decode_workspace_size(int64_t max_seq_len, int64_t batch, int64_t num_heads,
                                                int64_t num_kv_splits);
SGLK_API int sglk_flash_mla_decode(sglk_stream_t stream, void* out, const void* q_nope, const void* q_pe,
                                   const void* cache, const int32_t* seq_lens, const int32_t* page_table,
                                   void* workspace, int64_t workspace_bytes, int64_t batch,
                                   int64_t num_heads, int64_t page_size, int64_t pages_per_seq,
                                   int64_t q_nope_stride_b, int64_t q_nope_stride_h,
                                   int64_t q_pe_stride_b, int64_t q_pe_stride_h,
                                   int64_t cache_page_stride, int64_t table_stride, float sm_scale,
                                   int64_t num_kv_splits, int dtype);

/* moe_grouped_mm_nt_xe20 (16-bit weights): reference src/sycl/GroupGemmXe20.cpp:160-275.
 *   x = A_e @ W_e^T (+ bias_e fp32) for the rows of expert e; W [E, N, K] with row stride ldb and expert stride
 *   weight_stride_e (elements); A [total_m, K] contiguous; dtype BF16 / F16.
 *   fused_act 0: out [total_m, N] = T(x)
 *             1 / 2 (silu / tanh-gelu, gated: W = gate rows [0, N/2) then up rows): out [total_m, N/2] =
 *                   T(act(x[:, n]) * x[:, N/2 + n]) computed on the fp32 accumulators
 *             3 (relu2): out [total_m, N] = T(max(x, 0)^2)
 *   (reference kernels/moe/xe20/bf16/moe_mainloop.hpp:232-247, :375-390; common/activation.hpp:31-50) */
SGLK_API int sglk_moe_grouped_mm(sglk_stream_t stream, void* out, const void* activations, const void* weights,
                                 const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                 int64_t n_experts, int64_t N, int64_t K, int64_t ldb, int64_t weight_stride_e,
                                 int dtype, int fused_act);
/* moe_grouped_mm_nt_xe20 with activation_type 2 (swiglu_gpt_oss) and fuse_act = true: reference
 * kernels/moe/xe20/bf16/moe_kernel.hpp:109-125 (gate = weight rows 0, 2, 4, .., up = rows 1, 3, 5, .. - INTERLEAVED - and the
 * bias likewise, :138-146) and common/activation.hpp:31-42: out [total_m, N/2],
 *   out[m, n] = T(g * sigmoid(alpha * g) * (u + 1)),  g = min(x[m, 2n], limit),  u = clamp(x[m, 2n + 1], -limit, limit),
 * x = A W^T + bias on the fp32 accumulators. Other parameters as sglk_moe_grouped_mm. */
SGLK_API int sglk_moe_grouped_mm_swiglu(sglk_stream_t stream, void* out, const void* activations, const void* weights,
                                        const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                        int64_t n_experts, int64_t N, int64_t K, int64_t ldb, int64_t weight_stride_e,
                                        int dtype, float alpha, float limit);

/* ---- MLA prefill ------------------------------------------------------------
 * flash_mla_prefill: reference src/sycl/mla_prefill.cpp (schema src/torch_extension_sycl.cc:379-383;
 * wrapper python/sgl_kernel/attention.py:149-233; meaning tests/test_flash_mla_prefill.py:30-90).
 * Ragged queries q_nope [total_q, H, 512] / q_pe [total_q, H, 64] (strides in elements), sequence b owns
 * rows cu_seqlens_q[b] .. cu_seqlens_q[b+1]-1 and the first seq_lens_k[b] rows of its pages:
 *   out[t,h,:512] = softmax(sm_scale * [q_nope[t,h], q_pe[t,h]] . C_b[:n_t]^T) . C_b[:n_t, :512]
 * with n_t = seq_lens_k[b] - seqlen_q[b] + (t - cu_seqlens_q[b]) + 1 when causal (prefix unmasked,
 * bottom-right aligned), else seq_lens_k[b]. out is [total_q, H, 512] contiguous; exactly total_q rows are
 * written (no 256-row padding is needed, cf. attention.py:212-217). No workspace is used. */
SGLK_API int64_t sglk_flash_mla_prefill_workspace_size(int64_t max_seq_len, int64_t batch, int64_t num_heads,
                                                       int64_t page_size, int64_t num_kv_splits);
SGLK_API int sglk_flash_mla_prefill(sglk_stream_t stream, void* out, const void* q_nope, const void* q_pe,
                                    const void* cache, const int32_t* cu_seqlens_q, const int32_t* seq_lens_k,
                                    const int32_t* page_table, int64_t batch, int64_t max_seqlen_q,
                                    int64_t num_heads, int64_t page_size, int64_t pages_per_seq,
                                    int64_t q_nope_stride_t, int64_t q_nope_stride_h, int64_t q_pe_stride_t,
                                    int64_t q_pe_stride_h, int64_t cache_page_stride, int64_t table_stride,
                                    float sm_scale, int causal, int dtype);

/* ---- MoE routing, data movement and W4A16 grouped GEMM -----------------------
 * topk_softmax: reference src/sycl/TopKSoftMax.cpp:584-644 (schema torch_extension_sycl.cc:53).
 *   gating [tokens, experts] (F16/BF16/F32) -> topk_weights fp32 [tokens, k], topk_indices int32. */
SGLK_API int sglk_topk_softmax(sglk_stream_t stream, float* topk_weights, int32_t* topk_indices,
                               const void* gating, int64_t tokens, int64_t experts, int64_t topk,
                               int renormalize, int dtype);
/* moe_align_block_size: reference src/sycl/MoEAlign.cpp:313-383 (schema :199-203). num_experts
 * counts buckets (callers pass E+1: bucket = id + 1). sorted_token_ids / expert_ids /
 * num_tokens_post_pad / cumsum_buffer[num_experts + 1] are int32; ids I32 or I64. */
SGLK_API int sglk_moe_align_block_size(sglk_stream_t stream, const void* topk_ids, int ids_dtype,
                                       int64_t numel, int64_t num_experts, int64_t block_size,
                                       int32_t* sorted_token_ids, int32_t* expert_ids,
                                       int32_t* num_tokens_post_pad, int32_t* cumsum_buffer,
                                       int pad_sorted_token_ids);
/* prepare_moe_input: reference src/sycl/MoEPrepareInputs.cpp:459-497 (schema :219-223).
 * expert_counts[e] = rows of expert e; problem_sizes1/2 [E,3]; input_permutation[dst row] = token;
 * output_permutation[flat slot] = dst row. All index tensors share ids_dtype (I32 or I64). */
SGLK_API int sglk_prepare_moe_input(sglk_stream_t stream, const void* topk_ids, void* expert_counts,
                                    void* problem_sizes1, void* problem_sizes2,
                                    void* input_permutation, void* output_permutation, int64_t numel,
                                    int64_t topk, int64_t num_experts, int64_t n, int64_t k,
                                    int ids_dtype);
/* scatter_tokens_to_experts: reference src/sycl/MoEPrepareInputs.cpp:571-589 (schema :224).
 * output[src2dst_map[t*topk + j], :] = input[t, :]; rows of row_bytes bytes (multiple of 16). */
SGLK_API int sglk_scatter_tokens_to_experts(sglk_stream_t stream, const void* input,
                                            const int32_t* src2dst_map, void* output, int64_t tokens,
                                            int64_t topk, int64_t row_bytes);
/* apply_shuffle_mul_sum: reference src/sycl/MoEPrepareInputs.cpp:691-755 (schema :226-229).
 * output[t,:] = T( sum_j float(input[permutation[t*topk+j],:]) * factors[t,j] [* rsf] ); factors may be NULL. */
SGLK_API int sglk_apply_shuffle_mul_sum(sglk_stream_t stream, const void* input, void* output,
                                        const int32_t* permutation, const void* factors,
                                        int64_t tokens, int64_t topk, int64_t hidden,
                                        float routed_scaling_factor, int dtype, int factors_dtype);
/* moe_grouped_mm_nt_xe20_w4a16: reference src/sycl/GroupGemmW4A16Xe20.cpp:92-283 (schema
 * torch_extension_sycl.cc:214-217). out [total_m, N]; activations [total_m, K]; packed_weights
 * [E, N, K/2] (low nibble = even k); scales / zeros [E, N, K/group] (activation dtype; zeros may be
 * NULL = signed codes); bias fp32 [E, N] or NULL; rows_per_expert int32 [E] (counts). activations, packed_weights,
 * scales and zeros must be 16-byte aligned.
 * is_int4 == 0: mxfp4 weights (OCP e2m1 nibbles), scales = E8M0 bytes [E, N, K/32], group_size 32, zeros NULL
 * (reference GroupGemmW4A16Xe20.cpp:140-168, kernels/moe/xe20/w4a16/gemm_xe2.hpp:238-448). */
SGLK_API int sglk_moe_grouped_mm_w4a16(sglk_stream_t stream, void* out, const void* activations,
                                       const void* packed_weights, const void* scales,
                                       const void* zeros, const float* bias,
                                       const int32_t* rows_per_expert, int64_t total_m,
                                       int64_t n_experts, int64_t N, int64_t K, int64_t group_size,
                                       int is_int4, int dtype);
/* The same GEMM with the gate / up activation of fused_experts in its epilogue (reference python/sgl_kernel/moe.py:
 * 751-835 runs the GEMM, writes [rows, 2I] and calls silu_and_mul / gelu_tanh_and_mul on it; the 16-bit GEMM has the
 * fused form, kernels/moe/xe20/bf16/moe_mainloop.hpp:232-247). fused_act: 0 none, 1 silu, 2 gelu (tanh): W rows
 * [0, N/2) gate, [N/2, N) up, out [total_m, N/2] = T(act(gate + b) * (up + b)) from the fp32 accumulators;
 * 3 relu2: out [total_m, N] = T(max(x + b, 0)^2); 4 the DeepSeek-V4 clamped swiglu (reference silu_and_mul_clamp,
 * python/sgl_kernel/elementwise.py:231-255): gate = min(gate, act_limit), up = clamp(up, +-act_limit), silu(gate) * up.
 * row_map (may be NULL): int32 [total_m], the token gather of fused_experts (reference shuffle_rows,
 * python/sgl_kernel/moe.py:739) folded into the GEMM: row r of the expert-contiguous problem reads
 * activations[row_map[r]], activations is [src_rows, K] (src_rows * K < 2^32). A mapped call runs on the
 * streaming (decode) kernels at every size. */
SGLK_API int sglk_moe_grouped_mm_w4a16_act(sglk_stream_t stream, void* out, const void* activations,
                                           const void* packed_weights, const void* scales,
                                           const void* zeros, const float* bias,
                                           const int32_t* rows_per_expert, int64_t total_m,
                                           int64_t n_experts, int64_t N, int64_t K, int64_t group_size,
                                           int is_int4, int dtype, int fused_act, float act_limit,
                                           const int32_t* row_map, int64_t src_rows);

/* authored: the same gpt-oss swiglu epilogue on the 4-bit grouped GEMM (the reference runs its W4A16 GEMM 1 unfused and
 * calls swiglu_gpt_oss_sigmoid_alpha on the [rows, 2I] product, python/sgl_kernel/moe.py:748-789): gate = weight row 2n,
 * up = row 2n + 1, out [total_m, N/2]; other parameters as sglk_moe_grouped_mm_w4a16_act. */
SGLK_API int sglk_moe_grouped_mm_w4a16_swiglu(sglk_stream_t stream, void* out, const void* activations,
                                              const void* packed_weights, const void* scales, const void* zeros,
                                              const float* bias, const int32_t* rows_per_expert, int64_t total_m,
                                              int64_t n_experts, int64_t N, int64_t K, int64_t group_size, int is_int4,
                                              int dtype, float alpha, float limit, const int32_t* row_map,
                                              int64_t src_rows);
/* authored (no reference op): the DOWN projection of fused_experts (no bias, no activation) with the K range of every
 * 128 x 256 tile split over two workgroups, for the row counts at which that projection has fewer tiles than the GPU has CUs
 * (Mixtral: hidden 4096 = 16 column blocks x 8 row blocks of 224 K blocks at 512 tokens and again, with 256-row blocks, at
 * 1024). When the split is used (*split_used = the row block, 128 or 256) the rows of every full row block of an expert - and
 * of a remainder of more than half a block - are in ws = float [2][total_m][N] as two fp32 partial sums (their sum, rounded
 * once, is the GEMM's value), and only the rows of an expert's remainder of 1 .. block / 2 rows are in out;
 * sglk_apply_shuffle_mul_sum_splitk consumes both. When it is not used
 * (*split_used = 0: shape outside the regime, see sglk_moe_w4a16_splitk_applies) the call is sglk_moe_grouped_mm_w4a16
 * without a bias and ws is not touched. The reference picks a tile policy per average row count instead
 * (src/sycl/GroupGemmW4A16Xe20.cpp:266-277). */
SGLK_API int sglk_moe_grouped_mm_w4a16_splitk(sglk_stream_t stream, void* out, float* ws, const void* activations,
                                              const void* packed_weights, const void* scales, const void* zeros,
                                              const int32_t* rows_per_expert, int64_t total_m, int64_t n_experts,
                                              int64_t N, int64_t K, int64_t group_size, int is_int4, int dtype,
                                              int* split_used);
/* host-only: the row block (128 / 256) sglk_moe_grouped_mm_w4a16_splitk would split this shape with, 0 when it would not
 * (so that the caller can skip allocating ws) */
SGLK_API int sglk_moe_w4a16_splitk_applies(int64_t total_m, int64_t n_experts, int64_t N, int64_t K, int64_t group_size,
                                           int is_int4, int dtype);
/* apply_shuffle_mul_sum behind a split down projection: source row r is T(ws[0][r] + ws[1][r]) unless it lies in its
 * expert's remainder of 1 .. block_rows / 2 rows (from rows_per_expert [n_experts], expert-contiguous rows; block_rows =
 * *split_used of the GEMM call), then it is y[r]. */
SGLK_API int sglk_apply_shuffle_mul_sum_splitk(sglk_stream_t stream, const void* y, const float* ws, void* output,
                                               const int32_t* permutation, const void* factors,
                                               const int32_t* rows_per_expert, int64_t n_experts, int64_t block_rows,
                                               int64_t total_m,
                                               int64_t tokens, int64_t topk, int64_t hidden,
                                               float routed_scaling_factor, int dtype, int factors_dtype);

/* ---- flash-attention forward ---------------------------------------------------
 * fwd (mha_fwd): reference src/sycl/flash_attention.cpp:1332-1435 (schema
 * src/torch_extension_sycl.cc:328-358; wrappers python/sgl_kernel/flash_attn.py:103-372).
 *   q [total_q, Hq, D] ragged by cu_seqlens_q [b+1]; out same shape; lse fp32 [Hq, total_q].
 *   paged  (page_table != NULL): k/v [pages, page, Hk, D] with strides (page, token, head),
 *          seqlens_k = per-sequence lengths [b], page_table int32 [b, table_stride];
 *   ragged (page_table == NULL): k/v [total_k, Hk, D] with strides (token, head, unused),
 *          seqlens_k = cumulative [b+1].
 *   is_causal / window (left,right; < 0 = unlimited), bottom-right aligned; softcap 0 = off;
 *   sinks fp32 [Hq] or NULL; num_splits >= 1 (> 1 needs part_o fp32 [splits,total_q,Hq,D] and
 *   part_lse fp32 [splits,Hq,total_q]); sglk_attn_auto_splits gives the "0 = auto" choice: up to 64 packed rows
 *   per kv head (the decode kernel) about one workgroup per CU in splits of at least 4 tiles of 32 keys, at most eight of
 *   them unless a split would exceed 64 tiles, unsplit from 128 workgroups on below 4096 keys; above (128-row blocks) about
 *   two workgroups per CU in splits of at least 512 keys, unsplit from one workgroup per CU on below 16384 keys - the counts
 *   are pinned by tests/test_cabi.py.
 *   All k / v strides are in elements, non-negative and below 2^31. */
SGLK_API int64_t sglk_attn_auto_splits(int64_t batch, int64_t num_heads_k, int64_t max_rows_per_kv_head,
                                       int64_t max_seqlen_k);
/* kv_layout 0: ragged k / v [total_k, Hk, D] (strides token, head), seqlens_k = cumulative offsets [b + 1];
 *           1: paged [pages, page, Hk, D] + page_table; 2: one cache row per slot [slots, seqlen_cache, Hk, D]
 *              (strides slot, token, head; reference decode::mha_fwd_nopage, flash_attention.cpp:83-270).
 * Layouts 1 and 2: seqlens_k[b] is the END cache position of sequence b, its keys are the cache positions
 * [leftpad_k[b], seqlens_k[b]) (leftpad_k NULL: 0) of cache row kv_batch_idx[b] (NULL: b) - the page-table row or
 * the slot (reference flash_attention.cpp:383, :408-412, :649-653). */
/* kv_dtype: dtype (K/V stored like q) or SGLK_FP8_E4M3 / SGLK_FP8_E5M2 for an fp8 KV cache, dequantised in the kernel
 * with one float each for K and V (device pointers k_descale / v_descale; reference flash_attention.cpp:561-572). */
SGLK_API int sglk_attn_fwd(sglk_stream_t stream, void* out, float* lse, const void* q, const void* k,
                           const void* v, const int32_t* cu_seqlens_q, const int32_t* seqlens_k,
                           const int32_t* page_table, const float* sinks, float* part_o, float* part_lse,
                           int64_t batch, int64_t total_q, int64_t max_seqlen_q, int64_t num_heads,
                           int64_t num_heads_k, int64_t head_dim, int64_t page_size, int64_t q_stride0,
                           int64_t q_stride1, int64_t o_stride0, int64_t o_stride1, int64_t k_stride0,
                           int64_t k_stride1, int64_t k_stride2, int64_t v_stride0, int64_t v_stride1,
                           int64_t v_stride2, int64_t table_stride, float softmax_scale, int is_causal,
                           int64_t window_left, int64_t window_right, float softcap, int64_t num_splits,
                           int dtype, int kv_dtype, const float* k_descale, const float* v_descale, int kv_layout,
                           const int32_t* kv_batch_idx, const int32_t* leftpad_k);

/* sgl_per_token_group_quant_8bit_v2: reference src/sycl/per_token_group_quant_8bit_v2.cpp:714-842
 * (schema src/torch_extension_sycl.cc:399-402). As v1 plus: fuse_silu_and_mul (x is [.., 2*hidden], the value
 * quantised is T(T(silu(x1)) * x2)), and the expert-masked layout (x [experts, rows_per_expert, *], only rows
 * < masked_m[e] are processed; masked_m NULL = all rows, experts = 1). scale_kind as v1; float scales go to
 * e*s_stride_expert + row*s_stride_row + g*s_stride_col; packed ue8m0 to byte
 * (e*s_stride_expert + (g/4)*s_stride_col + row)*4 + g%4. hidden = OUTPUT hidden size. 16-bit inputs only. */
SGLK_API int sglk_per_token_group_quant_8bit_v2(sglk_stream_t stream, const void* x, void* q, void* scales,
                                                const int32_t* masked_m, int64_t num_experts,
                                                int64_t rows_per_expert, int64_t hidden, int group_size,
                                                float eps, float qmin, float qmax, int in_dtype,
                                                int out_dtype, int scale_kind, int64_t s_stride_expert,
                                                int64_t s_stride_row, int64_t s_stride_col,
                                                int fuse_silu_and_mul);

/* rotary_embedding: reference src/sycl/Rope.cpp:453-471 (schema torch_extension_sycl.cc:117-120).
 * In place when q_out == q (2-D form); out of place otherwise (3-D form). Strides in elements:
 * (token, head) for inputs and outputs. cos_sin_cache [max_pos, rot_dim] in the dtype of q. */
SGLK_API int sglk_rotary_embedding(sglk_stream_t stream, void* q_out, void* k_out, const void* q,
                                   const void* k, const int64_t* positions, const void* cos_sin_cache,
                                   int64_t tokens, int64_t num_heads, int64_t num_kv_heads,
                                   int64_t head_size, int64_t rot_dim, int64_t q_tok_stride,
                                   int64_t q_head_stride, int64_t k_tok_stride, int64_t k_head_stride,
                                   int64_t qo_tok_stride, int64_t qo_head_stride, int64_t ko_tok_stride,
                                   int64_t ko_head_stride, int is_neox, int dtype);

/* ---- attention prologue / epilogue ------------------------------------------------------
 * merge_state / merge_state_v2: reference src/sycl/merge_states.cpp:138-361 (schemas
 * src/torch_extension_sycl.cc:232-234). v_* [tokens, heads, head_size] contiguous, s_* [tokens, heads] fp32.
 *   m = max(s_a, s_b) (non-finite s counts as -inf); w_x = B^(s_x - m); z = max(w_a + w_b, FLT_MIN)
 *   v_merged = T(v_a w_a / z + v_b w_b / z);  s_merged = log_B(z) + m   (B = 2 when base2 != 0, else e)
 * s_merged may be NULL. dtype in {F32, F16, BF16}; head_size a multiple of 16 bytes of T. */
SGLK_API int sglk_merge_state(sglk_stream_t stream, void* v_merged, float* s_merged, const void* v_a,
                              const float* s_a, const void* v_b, const float* s_b, int64_t tokens, int64_t heads,
                              int64_t head_size, int dtype, int base2);

/* store_cache: reference src/sycl/KVCache.cpp:75-160 (schema src/torch_extension_sycl.cc:122-125).
 * Row t of k / v (row strides in BYTES; rows themselves contiguous) is copied to row indices[t] of the dense
 * k_cache / v_cache [cache_size, row_bytes]; indices[t] < 0 skips the token. Any element type (byte copy). */
SGLK_API int sglk_store_cache(sglk_stream_t stream, void* k_cache, void* v_cache, const void* k, const void* v,
                              const int64_t* indices, int64_t tokens, int64_t row_bytes, int64_t k_row_stride_bytes,
                              int64_t v_row_stride_bytes);

/* fused_inplace_qknorm_rope: reference src/sycl/FusedQKNormRope.cpp:1723-1861 (schema
 * src/torch_extension_sycl.cc:421-424). In place on q [tokens, Hq, D] / k [tokens, Hk, D] (strides in elements,
 * last dim contiguous): per head y = x * rsqrt(mean x^2 + eps) * w, then the first rope_dim elements are rotated
 * (neox or interleaved pairs) by the fp32 cos_sin_cache row of the token's position ([max_pos, rope_dim]: cos
 * first half, sin second half). D in {64, 128, 256}; positions int32 or int64; dtype in {F32, F16, BF16}. */
SGLK_API int sglk_fused_qknorm_rope_cache(sglk_stream_t stream, void* q, void* k, const void* q_weight,
                                          const void* k_weight, const float* cos_sin_cache, const void* positions,
                                          int positions_are_int64, int64_t tokens, int64_t num_q_heads,
                                          int64_t num_k_heads, int64_t head_dim, int64_t rope_dim,
                                          int64_t q_token_stride, int64_t q_head_stride, int64_t k_token_stride,
                                          int64_t k_head_stride, int is_neox, float eps, int dtype);

/* fused_qk_norm_rope: reference src/sycl/FusedQKNormRope.cpp:507-615 (schema src/torch_extension_sycl.cc:416-420).
 * Same arithmetic on a packed qkv [tokens, (Hq + Hk + Hv) * D] (V untouched), angles computed on the fly:
 * theta = position * freq(j), freq(j) = base^(-2j / rotary_dim) blended YaRN-style when factor != 1
 * (computeFreqYarn :42-67); the rotated elements are multiplied by attention_factor. positions int32. */
SGLK_API int sglk_fused_qknorm_rope_yarn(sglk_stream_t stream, void* qkv, const void* q_weight, const void* k_weight,
                                         const int32_t* position_ids, int64_t tokens, int64_t num_q_heads,
                                         int64_t num_k_heads, int64_t num_v_heads, int64_t head_dim,
                                         int64_t rotary_dim, float eps, float base, int is_neox, float factor,
                                         float low, float high, float attention_factor, int dtype);

/* ---- DeepSeek-style MoE routers -------------------------------------------------------
 * Experts are picked by iterative arg-max of choice = score (+ bias), ties -> the lower expert index; the weights
 * are the UNBIASED scores. gating / input [tokens, E] in {F32, F16, BF16}, E <= 512, topk <= 32; outputs fp32 /
 * int32 [tokens, topk]; fused shared-expert slot i gets id E + i.
 * topk_sigmoid: reference src/sycl/TopKSigMoid.cpp (schema src/torch_extension_sycl.cc:55-58). */
SGLK_API int sglk_topk_sigmoid(sglk_stream_t stream, float* topk_weights, int32_t* topk_ids, const void* gating,
                               const float* correction_bias, int64_t tokens, int64_t num_experts, int64_t topk,
                               int renormalize, float routed_scaling_factor, int64_t num_fused_shared_experts,
                               int dtype);
/* biased_topk: reference src/sycl/BiasedTopK.cpp (schema src/torch_extension_sycl.cc:111-115);
 * scoring_func 0 = sigmoid, 1 = sqrt(softplus). */
SGLK_API int sglk_biased_topk(sglk_stream_t stream, float* output, int32_t* indices, const void* input,
                              const float* bias, int64_t tokens, int64_t num_experts, int64_t topk, int scoring_func,
                              int64_t num_fused_shared_experts, int renormalize, float routed_scaling_factor,
                              int apply_routed_scaling_factor_on_output, int dtype);
/* moe_fused_gate: reference src/sycl/MoE_fused_gate.cpp (schema src/torch_extension_sycl.cc:191-196); grouped
 * top-k (groups ranked by the sum of their two largest choices; softmax scoring: by the largest);
 * scoring_func 0 = sigmoid, 1 = softmax; bias (may be NULL) in the dtype of input. */
SGLK_API int sglk_moe_fused_gate(sglk_stream_t stream, float* output, int32_t* indices, const void* input,
                                 const void* bias, int64_t tokens, int64_t num_experts, int64_t num_expert_group,
                                 int64_t topk_group, int64_t topk, int64_t num_fused_shared_experts, int scoring_func,
                                 int renormalize, float routed_scaling_factor,
                                 int apply_routed_scaling_factor_on_output, int dtype);

/* ---- top-k / top-p / min-p filtering and sampling -------------------------------------------
 * reference src/sycl/TopKRenormProbs.cpp, TopPRenormProbs.cpp, TopKTopPSamplingFromProbs.cpp,
 * MinPSamplingFromProbs.cpp (schemas src/torch_extension_sycl.cc:66-80). probs fp32 [rows, vocab] contiguous.
 *   kept(top-k): x >= k-th largest value; kept(top-p): x >= t_p, t_p the largest t with mass{x >= t} >= p;
 *   kept(min-p): x >= min_p * max(x). Pivots are exact (radix select on the float bits, fixed-point masses).
 *   renorm: out = kept ? x / sum(kept) : 0.  sampling: index drawn ~ x over the kept set by inverse CDF of one
 *   Philox4x32-10(seed; offset, row) 64-bit draw: reproducible from (seed, offset). Per-row parameter arrays are
 *   optional (NULL: the scalar); indices (optional) maps output row b to probs row indices[b]. */
SGLK_API int sglk_top_k_renorm_probs(sglk_stream_t stream, float* renorm_probs, const float* probs,
                                     const int64_t* top_k_arr, int64_t top_k_val, int64_t batch, int64_t vocab);
SGLK_API int sglk_top_p_renorm_probs(sglk_stream_t stream, float* renorm_probs, const float* probs,
                                     const float* top_p_arr, float top_p_val, int64_t batch, int64_t vocab);
SGLK_API int sglk_top_k_top_p_sampling_from_probs(sglk_stream_t stream, int32_t* output, const float* probs,
                                                  const int64_t* indices, const int32_t* top_k_arr, int64_t top_k_val,
                                                  const float* top_p_arr, float top_p_val, int use_top_k,
                                                  int64_t batch, int64_t vocab, uint64_t philox_seed,
                                                  uint64_t philox_offset);
SGLK_API int sglk_min_p_sampling_from_probs(sglk_stream_t stream, int32_t* output, const float* probs,
                                            const int64_t* indices, const float* min_p_arr, float min_p_val,
                                            int64_t batch, int64_t vocab, uint64_t philox_seed,
                                            uint64_t philox_offset);
/* The same two draws for a launch RECORDED INTO A HIP GRAPH: the generator state lives on the device (what torch's
 * PhiloxCudaState carries while a stream is capturing) - seed = *philox_seed_ptr, offset = *philox_offset_ptr +
 * offset_intragraph, read by the kernel at replay, so that every replay draws fresh numbers. The reference reads its
 * generator on the host at each call (TopKTopPSamplingFromProbs.cpp / MinPSamplingFromProbs.cpp:270-275) and has no
 * counterpart; results equal the scalar entries' at the same (seed, offset). */
SGLK_API int sglk_top_k_top_p_sampling_from_probs_graph(sglk_stream_t stream, int32_t* output, const float* probs,
                                                        const int64_t* indices, const int32_t* top_k_arr,
                                                        int64_t top_k_val, const float* top_p_arr, float top_p_val,
                                                        int use_top_k, int64_t batch, int64_t vocab,
                                                        const int64_t* philox_seed_ptr, const int64_t* philox_offset_ptr,
                                                        uint64_t offset_intragraph);
SGLK_API int sglk_min_p_sampling_from_probs_graph(sglk_stream_t stream, int32_t* output, const float* probs,
                                                  const int64_t* indices, const float* min_p_arr, float min_p_val,
                                                  int64_t batch, int64_t vocab, const int64_t* philox_seed_ptr,
                                                  const int64_t* philox_offset_ptr, uint64_t offset_intragraph);
/* The five ops above for a decode batch over a long vocabulary, with a scratch buffer the caller owns (no counterpart in the
 * reference): up to 64 rows of at least 32768 entries are cut into several workgroups per row, the select's passes become
 * launches and the rows' histograms meet in the workspace - the same bits as the entries above (every sum is an integer),
 * ~3x sooner at batch 1 .. 8. op: 0 top_k_renorm_probs, 1 top_p_renorm_probs, 2 top_k_top_p_sampling_from_probs (both
 * filters), 3 top_p_sampling_from_probs, 4 min_p_sampling_from_probs; result = renorm_probs (float) or output (int32);
 * top_k_arr int64 (top_k_is_int64 = 1, the renorm op) or int32; p_arr / p_val = top-p or min-p; the generator state by value, or
 * by pointer (both non-NULL: a launch recorded into a HIP graph, philox_offset then is the offset inside the graph).
 * _workspace_size: bytes (0: one workgroup per row, whatever is passed); a NULL or short workspace is not an error. */
SGLK_API int64_t sglk_sampling_workspace_size(int64_t batch, int64_t vocab);
SGLK_API int sglk_sampling_ws(sglk_stream_t stream, int op, void* result, const float* probs, const int64_t* indices,
                              const void* top_k_arr, int top_k_is_int64, int64_t top_k_val, const float* p_arr,
                              float p_val, int64_t batch, int64_t vocab, uint64_t philox_seed, uint64_t philox_offset,
                              const int64_t* philox_seed_ptr, const int64_t* philox_offset_ptr, void* workspace,
                              int64_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SGLK_H_ */
