"""MoE routing, data movement, W4A16 grouped GEMM and fused_experts — CPU restatements.

  topk_softmax          reference src/sycl/TopKSoftMax.cpp:249-476: softmax in fp32, k rounds of arg-max with
                        strict '>' (lower index wins ties), weights = probabilities (x 1/sum when
                        renormalize). The arg-max is taken on the logits: the same order as on the
                        probabilities (exp is monotone) but free of underflow/rounding ties, so GPU and CPU
                        agree index for index. Against the reference's own torch.topk-on-softmax vectors the
                        comparison is modulo equal-score ties, exactly like tests/test_topk_softmax.py:12-37.
  moe_align_block_size  reference src/sycl/MoEAlign.cpp:60-214 (bucket = id + 1, padded counts, exclusive scan,
                        expert_ids = bucket - 1) and :35-58 (placement; order inside a bucket unspecified).
                        numpy, exact.
  prepare_moe_input     reference src/sycl/MoEPrepareInputs.cpp:36-61, :207-240 (counts, problem sizes, a_map,
                        c_map) with the stable (flat slot) order inside an expert. numpy, exact.
  w4a16 grouped GEMM    reference src/sycl/kernels/moe/xe20/w4a16/gemm_xe2.hpp:405-428 + :52-76: every weight is
                        T((code - zp) * scale) (one rounding to the activation dtype), fp32 accumulation,
                        fp32 bias, one rounding of the output. Packing per tests/test_moe_gemm.py:293-295.
  fused_experts         reference python/sgl_kernel/moe.py:403-870 op sequence on top of the pieces above;
                        pinned against torch_naive_moe (tests/test_moe_gemm.py:59-138) golden vectors.
"""
import numpy as np
import torch

from oracle import activation as oact


def topk_softmax(gating: torch.Tensor, topk: int, renormalize: bool):
    p = torch.softmax(gating.float(), dim=-1)
    work = gating.float().clone()
    T, E = p.shape
    w = torch.empty(T, topk, dtype=torch.float32)
    idx = torch.empty(T, topk, dtype=torch.int32)
    ar = torch.arange(T)
    for j in range(topk):
        m = work.max(dim=-1, keepdim=True).values
        first = (work == m).float().argmax(dim=-1)  # lowest index among the maxima
        w[:, j] = p[ar, first]
        idx[:, j] = first.to(torch.int32)
        work[ar, first] = float("-inf")
    if renormalize:
        w = w * (1.0 / w.sum(dim=-1, keepdim=True))
    return w, idx, p


def moe_align_block_size(topk_ids: np.ndarray, num_experts: int, block_size: int):
    """Returns (sorted_token_ids (stable order), expert_ids, num_tokens_post_pad, prefix) for bucket = id + 1."""
    flat = np.asarray(topk_ids).reshape(-1).astype(np.int64) + 1
    numel = flat.size
    counts = np.bincount(flat[(flat >= 0) & (flat < num_experts)], minlength=num_experts)
    padded = (counts + block_size - 1) // block_size * block_size
    prefix = np.concatenate([[0], np.cumsum(padded)]).astype(np.int32)
    total = int(prefix[-1])
    sorted_ids = np.full(max(total, 1), numel, dtype=np.int32)[:total]
    for b in range(num_experts):
        where = np.nonzero(flat == b)[0]
        sorted_ids[prefix[b]:prefix[b] + where.size] = where
    expert_ids = np.empty(total // block_size, dtype=np.int32)
    for i in range(total // block_size):
        expert_ids[i] = np.searchsorted(prefix[:num_experts], i * block_size, side="right") - 2
    return sorted_ids, expert_ids, total, prefix


def prepare_moe_input(topk_ids: np.ndarray, num_experts: int, n: int, k: int):
    ids = np.asarray(topk_ids)
    topk = ids.shape[1]
    flat = ids.reshape(-1)
    counts = np.bincount(flat[flat >= 0], minlength=num_experts)[:num_experts].astype(np.int32)
    order = np.argsort(flat, kind="stable")
    order = order[flat[order] >= 0]
    a_map = (order // topk).astype(np.int32)          # dst row -> src token
    c_map = np.empty(flat.size, dtype=np.int32)       # flat slot -> dst row
    c_map[order] = np.arange(order.size, dtype=np.int32)
    ps1 = np.stack([counts, np.full_like(counts, 2 * n), np.full_like(counts, k)], axis=1)
    ps2 = np.stack([counts, np.full_like(counts, k), np.full_like(counts, n)], axis=1)
    return counts, ps1, ps2, a_map, c_map


def unpack_int4(packed: torch.Tensor, signed: bool) -> torch.Tensor:
    """[.., K/2] bytes -> [.., K] integer codes (low nibble = even k)."""
    b = packed.view(torch.uint8).to(torch.int16)
    lo, hi = b & 0xF, (b >> 4) & 0xF
    codes = torch.stack([lo, hi], dim=-1).reshape(*packed.shape[:-1], packed.shape[-1] * 2)
    if signed:
        codes = torch.where(codes >= 8, codes - 16, codes)
    return codes


def dequant_w4(packed, scales, zeros, group_size):
    """T((code - zp) * scale): the reference kernel's per-weight rounding."""
    T = scales.dtype
    codes = unpack_int4(packed, signed=zeros is None).float()
    s = scales.float().repeat_interleave(group_size, dim=-1)
    if zeros is not None:
        z = zeros.float().repeat_interleave(group_size, dim=-1)
        v = (codes - z).to(T).float()
    else:
        v = codes.to(T).float()
    return (v * s).to(T)


_E2M1 = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)  # OCP MX e2m1 magnitudes by the low three code bits


def dequant_mxfp4(packed, scales, dtype):
    """T(T(e2m1 value) * 2^(scale byte - 127)) with one E8M0 byte per 32 weights: the dequantisation the reference's
    tests pair with the mxfp4 op (tests/test_per_token_group_quant_mxfp4.py:175-288, used by tests/test_moe_gemm.py:
    273-290); the kernel side is src/sycl/GroupGemmW4A16Xe20.cpp:140-168."""
    codes = unpack_int4(packed, signed=False).long()
    mag = torch.tensor(_E2M1, dtype=torch.float32)[codes & 7]
    v = torch.where((codes & 8) != 0, -mag, mag).to(dtype).float()
    exp = scales.view(torch.uint8).to(torch.int32) - 127
    s = torch.pow(2.0, exp.float()).repeat_interleave(32, dim=-1)
    return (v * s).to(dtype)


def moe_grouped_mm_w4a16(act, packed, scales, zeros, bias, rows_per_expert, group_size, mxfp4=False):
    T = act.dtype
    w = dequant_mxfp4(packed, scales, T) if mxfp4 else dequant_w4(packed, scales, zeros, group_size)  # [E, N, K]
    out = torch.empty(act.shape[0], w.shape[1], dtype=T)
    r0 = 0
    for e, r in enumerate(rows_per_expert.tolist()):
        if r:
            o = act[r0:r0 + r].float() @ w[e].float().t()
            if bias is not None:
                o = o + bias[e].float()
            out[r0:r0 + r] = o.to(T)
        r0 += r
    return out


def fused_experts_int4(x, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale, w1_zp=None, w2_zp=None, b1=None,
                       b2=None, activation="silu", routed_scaling_factor=None, mxfp4=False, gemm1_alpha=None,
                       gemm1_limit=None):
    T = x.dtype
    E = w1.shape[0]
    K = w1.shape[2] * 2
    I = w2.shape[2] * 2
    topk = topk_ids.shape[1]
    counts, _, _, a_map, c_map = prepare_moe_input(topk_ids.numpy(), E, x.shape[1], topk)
    a = x[torch.from_numpy(a_map).long()]
    rows = torch.from_numpy(counts)
    h = moe_grouped_mm_w4a16(a, w1, w1_scale, w1_zp, b1.float() if b1 is not None else None, rows, K // w1_scale.shape[2],
                             mxfp4)
    if gemm1_alpha is not None:
        h = oact.swiglu_gpt_oss_sigmoid_alpha(h, gemm1_alpha, gemm1_limit)
    elif activation == "silu":
        h = oact.silu_and_mul(h)
    elif activation == "gelu":
        h = oact.gelu_tanh_and_mul(h)
    else:
        h = torch.square(torch.relu(h))
    o = moe_grouped_mm_w4a16(h, w2, w2_scale, w2_zp, b2.float() if b2 is not None else None, rows, I // w2_scale.shape[2],
                             mxfp4)
    gathered = o[torch.from_numpy(c_map).long()].view(x.shape[0], topk, -1).float()
    t = gathered * topk_weights.float().unsqueeze(-1)
    if routed_scaling_factor is not None and routed_scaling_factor != 1.0:
        t = t * routed_scaling_factor
    acc = torch.zeros(x.shape[0], o.shape[1])
    for j in range(topk):  # slot order, fp32
        acc = acc + t[:, j]
    return acc.to(T)


def moe_grouped_mm(act, weights, bias, rows_per_expert):
    """moe_grouped_mm_nt_xe20 without the fused epilogue (reference src/sycl/GroupGemmXe20.cpp:160-275): expert e
    multiplies its rows by W_e^T (+ fp32 bias), fp32 accumulation, one rounding to the activation dtype."""
    T = act.dtype
    out = torch.empty(act.shape[0], weights.shape[1], dtype=T)
    r0 = 0
    for e, r in enumerate(rows_per_expert.tolist()):
        if r:
            o = act[r0:r0 + r].float() @ weights[e].float().t()
            if bias is not None:
                o = o + bias[e].float()
            out[r0:r0 + r] = o.to(T)
        r0 += r
    return out


def moe_grouped_mm_fused(act, weights, bias, rows_per_expert, activation, alpha=1.702, limit=7.0):
    """moe_grouped_mm_nt_xe20 with fuse_act (reference kernels/moe/xe20/bf16/moe_mainloop.hpp:232-247, :375-390,
    common/activation.hpp:31-50): the activation works on the fp32 accumulators (+ bias), one rounding to T.
    silu / gelu: weights hold gate rows then up rows, out [rows, N/2]; relu2: out [rows, N] = max(x, 0)^2;
    swiglu_gpt_oss: gate = weight rows 0, 2, 4, .., up = rows 1, 3, 5, .. (moe_kernel.hpp:109-125; bias likewise, :138-146),
    out = min(gate, limit) * sigmoid(alpha * that) * (clamp(up, +-limit) + 1) (common/activation.hpp:35-41)."""
    T = act.dtype
    n = weights.shape[1]
    out = torch.empty(act.shape[0], n if activation == "relu2" else n // 2, dtype=T)
    r0 = 0
    for e, r in enumerate(rows_per_expert.tolist()):
        if r:
            o = act[r0:r0 + r].float() @ weights[e].float().t()
            if bias is not None:
                o = o + bias[e].float()
            if activation == "relu2":
                o = torch.square(torch.relu(o))
            elif activation == "swiglu_gpt_oss":
                gt, up = o[:, 0::2].clamp(max=limit), o[:, 1::2].clamp(min=-limit, max=limit)
                o = gt * (1.0 / (1.0 + torch.exp(-(gt * alpha)))) * (up + 1.0)
            else:
                g, u = o[:, : n // 2], o[:, n // 2:]
                if activation == "silu":
                    o = g * torch.sigmoid(g) * u
                else:
                    o = g * (0.5 * (1.0 + torch.tanh(0.7978845608028654 * (g + 0.044715 * g * g * g)))) * u
            out[r0:r0 + r] = o.to(T)
        r0 += r
    return out


def fused_experts_16bit(x, w1, w2, topk_weights, topk_ids, b1=None, b2=None, activation="silu", routed_scaling_factor=None,
                        fused_epilogue=False, gemm1_alpha=None, gemm1_limit=None):
    """fused_experts with 16-bit weights: the op sequence of reference python/sgl_kernel/moe.py:742-866 (GEMM1, gated
    activation in the activation dtype, GEMM2, fp32 weighted combine in slot order); same result as torch_naive_moe
    (tests/test_moe_gemm.py:59-137) up to the rounding of the two intermediates."""
    T = x.dtype
    E = w1.shape[0]
    topk = topk_ids.shape[1]
    counts, _, _, a_map, c_map = prepare_moe_input(topk_ids.numpy(), E, x.shape[1], topk)
    a = x[torch.from_numpy(a_map).long()]
    rows = torch.from_numpy(counts)
    if gemm1_alpha is not None:  # gpt-oss swiglu (moe.py:692-697, :787-789): GEMM 1 rounded to T, interleaved gate / up
        h = moe_grouped_mm(a, w1, b1.float() if b1 is not None else None, rows)
        h = oact.swiglu_gpt_oss_sigmoid_alpha(h, gemm1_alpha, gemm1_limit)
    elif fused_epilogue:  # the reference's fuse_act route (moe.py:812-860): no rounding between GEMM 1 and the activation
        h = moe_grouped_mm_fused(a, w1, b1.float() if b1 is not None else None, rows, activation)
    else:
        h = moe_grouped_mm(a, w1, b1.float() if b1 is not None else None, rows)
        if activation == "silu":
            h = oact.silu_and_mul(h)
        elif activation == "gelu":
            h = oact.gelu_tanh_and_mul(h)
        else:
            h = torch.square(torch.relu(h))
    o = moe_grouped_mm(h, w2, b2.float() if b2 is not None else None, rows)
    gathered = o[torch.from_numpy(c_map).long()].view(x.shape[0], topk, -1).float()
    t = gathered * topk_weights.float().unsqueeze(-1)
    if routed_scaling_factor is not None and routed_scaling_factor != 1.0:
        t = t * routed_scaling_factor
    acc = torch.zeros(x.shape[0], o.shape[1])
    for j in range(topk):  # slot order, fp32
        acc = acc + t[:, j]
    return acc.to(T)
