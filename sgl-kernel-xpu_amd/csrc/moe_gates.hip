// DeepSeek-style MoE routers: topk_sigmoid, biased_topk, moe_fused_gate.
//
// Replace reference src/sycl/TopKSigMoid.cpp (schema src/torch_extension_sycl.cc:55-58), src/sycl/BiasedTopK.cpp
// (:111-115) and src/sycl/MoE_fused_gate.cpp (:191-196). All three pick experts by iterative arg-max of a
// "choice" score (ties -> the LOWER expert index, as the reference's reductions: BiasedTopK.cpp:131-136,
// MoE_fused_gate.cpp:256-260) and report the UNBIASED score of the picked experts as routing weights:
//   topk_sigmoid   : score = sigmoid(x); choice = score + correction_bias (optional)
//                    renormalize: w *= rsf / (sum + 1e-20); one fused shared slot: id E, weight renorm ? 1 : sum / rsf
//                    (TopKSigMoid.cpp:96-176)
//   biased_topk    : score = sigmoid(x) or sqrt(softplus(x)); choice = score + bias;
//                    w_out = (w / (renorm && sum > 0 ? sum : 1)) * (apply ? rsf : 1); shared slots i: id E + i,
//                    weight sum / rsf before that normalisation (BiasedTopK.cpp:100-170)
//   moe_fused_gate : score = sigmoid(x) or softmax(x); choice = score + bias (optional); expert groups of E / G
//                    consecutive experts are ranked by the sum of their two largest choices (softmax: the largest),
//                    ties -> the lower group; only experts of the topk_group best groups can be picked;
//                    renormalize: w *= 1 / sum (0 when sum <= 0), then *= rsf when apply; shared slots i: id E + i,
//                    weight sum / rsf (MoE_fused_gate.cpp:130-330)
// Index paths are exact integer work: every comparison is on fp32 values computed the same way for every candidate.
//
// Kernel: one wave per token (4 tokens per workgroup), expert e lives in lane e % 64, register e / 64 (E <= 512);
// scores are parked in LDS for the gather of the picked weights; an arg-max step is a local scan + 6 xor-shuffles.
// Latency-bound (the input is T x E values); no attempt at anything else.
#include <float.h>
#include <math.h>

#include "common.h"

namespace sglk {
namespace {

constexpr int kMaxE = 512, kVPL = kMaxE / 64, kMaxTopK = 32;
enum { GATE_TOPK_SIGMOID = 0, GATE_BIASED = 1, GATE_GROUPED = 2 };
enum { SCORE_SIGMOID = 0, SCORE_SQRTSOFTPLUS = 1, SCORE_SOFTMAX = 2 };

struct GateParams {
  float* weights;   // [T, topk]
  int32_t* ids;     // [T, topk]
  const void* x;    // [T, E]
  const void* bias; // [E] fp32 (topk_sigmoid, biased_topk) or T (moe_fused_gate); may be null
  int64_t tokens;
  int E, topk, shared, scoring, renorm, apply_scale, groups, topk_group, bias_is_f32;
  float rsf;
};

__device__ __forceinline__ void better(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void gate_kernel(GateParams p) {
  __shared__ float s_score[4][kMaxE];
  __shared__ float s_choice[MODE == GATE_GROUPED ? 4 : 1][MODE == GATE_GROUPED ? kMaxE : 1];  // score + bias (the group scan's operand)
  __shared__ float s_group[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= p.tokens) return;
  const int E = p.E;
  const T* x = reinterpret_cast<const T*>(p.x) + row * E;
  float* sc = s_score[wave];

  float score[kVPL], choice[kVPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < kVPL; ++i) {
    const int e = lane + 64 * i;
    score[i] = e < E ? (float)x[e] : -INFINITY;
    mx = fmaxf(mx, score[i]);
  }
  float denom = 1.f;
  if (p.scoring == SCORE_SOFTMAX) {
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kVPL; ++i) s += (lane + 64 * i < E) ? expf(score[i] - mx) : 0.f;
    denom = wave_sum(s);
  }
#pragma unroll
  for (int i = 0; i < kVPL; ++i) {
    const int e = lane + 64 * i;
    float s = 0.f, c = -INFINITY;
    if (e < E) {
      const float v = score[i];
      s = p.scoring == SCORE_SIGMOID ? 1.0f / (1.0f + expf(-v))
          : p.scoring == SCORE_SQRTSOFTPLUS ? sqrtf(log1pf(expf(v)))
                                            : expf(v - mx) / denom;
      float b = 0.f;
      if (p.bias != nullptr) b = p.bias_is_f32 ? reinterpret_cast<const float*>(p.bias)[e] : (float)reinterpret_cast<const T*>(p.bias)[e];
      c = s + b;
      sc[e] = s;
      if constexpr (MODE == GATE_GROUPED) s_choice[wave][e] = c;
    }
    score[i] = s;
    choice[i] = c;
  }

  if constexpr (MODE == GATE_GROUPED) {
    // group scores: group g = experts [g * gs, (g + 1) * gs); lane g scans its group in LDS order
    const int G = p.groups, gs = E / G;
    float* sg = s_group[wave];
    __builtin_amdgcn_wave_barrier();
    float gscore = -INFINITY;
    if (lane < G) {
      float m1 = -INFINITY, m2 = -INFINITY;
      // (the biased scores from LDS: this loop used to fetch its gs bias values from global memory one after the other - most of
      //  the op's 15 us at decode)
      for (int j = 0; j < gs; ++j) {
        const float c = s_choice[wave][lane * gs + j];
        if (c > m1) { m2 = m1; m1 = c; } else if (c > m2) { m2 = c; }
      }
      gscore = p.scoring == SCORE_SOFTMAX ? m1 : m1 + m2;
      sg[lane] = gscore;
    }
    __builtin_amdgcn_wave_barrier();
    // rank of my experts' groups: kept iff fewer than topk_group groups are better (higher score, or equal and lower index)
#pragma unroll
    for (int i = 0; i < kVPL; ++i) {
      const int e = lane + 64 * i;
      if (e < E) {
        const int g = e / gs;
        const float mine = sg[g];
        int rank = 0;
        for (int o = 0; o < G; ++o) {
          const float ov = sg[o];
          rank += (ov > mine || (ov == mine && o < g)) ? 1 : 0;
        }
        if (rank >= p.topk_group) choice[i] = -INFINITY;
      }
    }
  } else {
    __builtin_amdgcn_wave_barrier();
  }

  const int routed = p.topk - p.shared;
  float my_w = 0.f;
  int my_id = 0;
  for (int k = 0; k < routed; ++k) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < kVPL; ++i) {
      const int e = lane + 64 * i;
      if (e < E && (choice[i] > bv)) { bv = choice[i]; bi = e; }  // ascending e per lane: strict > keeps the lower index
    }
    if (bi == 0x7fffffff && lane < E) bi = lane;  // (every candidate is -inf: fall back to an in-range index)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) better(bv, bi, __shfl_xor(bv, off, 64), __shfl_xor(bi, off, 64));
    if (bi >= E) bi = 0;
    const float w = sc[bi];
    if (lane == k) { my_w = w; my_id = bi; }
#pragma unroll
    for (int i = 0; i < kVPL; ++i)
      if (lane + 64 * i == bi) choice[i] = -INFINITY;
  }
  const float sum = wave_sum(lane < routed ? my_w : 0.f);

  if (lane < p.topk) {
    float w = my_w;
    int id = my_id;
    const bool is_shared = lane >= routed;
    if constexpr (MODE == GATE_TOPK_SIGMOID) {
      if (is_shared) { id = E + (lane - routed); w = p.renorm ? 1.0f : sum / p.rsf; }
      else if (p.renorm) w = w * (p.rsf / (sum + 1e-20f));
    } else if constexpr (MODE == GATE_BIASED) {
      if (is_shared) { id = E + (lane - routed); w = sum / p.rsf; }
      const float norm = (p.renorm && sum > 0.f) ? sum : 1.0f;
      w = (w / norm) * (p.apply_scale ? p.rsf : 1.0f);
    } else {
      if (is_shared) { id = E + (lane - routed); w = sum / p.rsf; }
      if (p.renorm) {
        w = w * (sum > 0.f ? 1.0f / sum : 0.f);
        if (p.apply_scale) w *= p.rsf;
      }
    }
    p.weights[row * p.topk + lane] = w;
    p.ids[row * p.topk + lane] = id;
  }
}

template <int MODE>
static int launch(hipStream_t st, const GateParams& p, int dtype, const char* op) {
  SGLK_REQUIRE(p.E > 0 && p.E <= kMaxE, "%s: num_experts must be in [1, %d], got %d", op, kMaxE, p.E);
  SGLK_REQUIRE(p.topk > p.shared && p.shared >= 0, "%s: topk must be greater than num_fused_shared_experts", op);
  SGLK_REQUIRE(p.topk <= kMaxTopK && p.topk - p.shared <= p.E, "%s: topk exceeds maximum supported value: %d", op, kMaxTopK);
  if (p.tokens == 0) return SGLK_OK;
  const unsigned grid = (unsigned)cdiv(p.tokens, 4);
  SGLK_DISPATCH_FLOAT(dtype, T, (gate_kernel<T, MODE><<<grid, 256, 0, st>>>(p)))
  return check_launch(op);
}

}  // namespace
}  // namespace sglk

extern "C" int sglk_topk_sigmoid(sglk_stream_t stream, float* topk_weights, int32_t* topk_ids, const void* gating,
                                 const float* correction_bias, int64_t tokens, int64_t num_experts, int64_t topk,
                                 int renormalize, float routed_scaling_factor, int64_t num_fused_shared_experts, int dtype) {
  using namespace sglk;
  GateParams p{};
  p.weights = topk_weights; p.ids = topk_ids; p.x = gating; p.bias = correction_bias; p.bias_is_f32 = 1;
  p.tokens = tokens; p.E = (int)num_experts; p.topk = (int)topk; p.shared = (int)num_fused_shared_experts;
  p.scoring = SCORE_SIGMOID; p.renorm = renormalize; p.rsf = routed_scaling_factor;
  return launch<GATE_TOPK_SIGMOID>((hipStream_t)stream, p, dtype, "topk_sigmoid");
}

extern "C" int sglk_biased_topk(sglk_stream_t stream, float* output, int32_t* indices, const void* input, const float* bias,
                                int64_t tokens, int64_t num_experts, int64_t topk, int scoring_func,
                                int64_t num_fused_shared_experts, int renormalize, float routed_scaling_factor,
                                int apply_routed_scaling_factor_on_output, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(scoring_func == 0 || scoring_func == 1, "scoring_func must be 0 (sigmoid) or 1 (sqrtsoftplus)");
  GateParams p{};
  p.weights = output; p.ids = indices; p.x = input; p.bias = bias; p.bias_is_f32 = 1;
  p.tokens = tokens; p.E = (int)num_experts; p.topk = (int)topk; p.shared = (int)num_fused_shared_experts;
  p.scoring = scoring_func == 0 ? SCORE_SIGMOID : SCORE_SQRTSOFTPLUS; p.renorm = renormalize;
  p.rsf = routed_scaling_factor; p.apply_scale = apply_routed_scaling_factor_on_output;
  return launch<GATE_BIASED>((hipStream_t)stream, p, dtype, "biased_topk");
}

extern "C" int sglk_moe_fused_gate(sglk_stream_t stream, float* output, int32_t* indices, const void* input, const void* bias,
                                   int64_t tokens, int64_t num_experts, int64_t num_expert_group, int64_t topk_group,
                                   int64_t topk, int64_t num_fused_shared_experts, int scoring_func, int renormalize,
                                   float routed_scaling_factor, int apply_routed_scaling_factor_on_output, int dtype) {
  using namespace sglk;
  SGLK_REQUIRE(scoring_func == 0 || scoring_func == 1, "scoring_func must be 0 (sigmoid) or 1 (softmax), but got %d", scoring_func);
  SGLK_REQUIRE(num_expert_group > 0 && num_expert_group <= 64 && num_experts % num_expert_group == 0,
               "num_experts must be divisible by num_expert_group (at most 64 groups), but got %lld / %lld",
               (long long)num_experts, (long long)num_expert_group);
  SGLK_REQUIRE(topk_group > 0 && topk_group <= num_expert_group, "moe_fused_gate: topk_group must be in [1, num_expert_group]");
  SGLK_REQUIRE((topk - num_fused_shared_experts) <= topk_group * (num_experts / num_expert_group),
               "moe_fused_gate: topk exceeds the experts of the selected groups");
  GateParams p{};
  p.weights = output; p.ids = indices; p.x = input; p.bias = bias; p.bias_is_f32 = 0;
  p.tokens = tokens; p.E = (int)num_experts; p.topk = (int)topk; p.shared = (int)num_fused_shared_experts;
  p.scoring = scoring_func == 0 ? SCORE_SIGMOID : SCORE_SOFTMAX; p.renorm = renormalize; p.rsf = routed_scaling_factor;
  p.apply_scale = apply_routed_scaling_factor_on_output; p.groups = (int)num_expert_group; p.topk_group = (int)topk_group;
  return launch<GATE_GROUPED>((hipStream_t)stream, p, dtype, "moe_fused_gate");
}
