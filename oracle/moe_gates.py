"""TEST INFRASTRUCTURE (CPU oracle) — DeepSeek-style MoE routers: topk_sigmoid, biased_topk, moe_fused_gate.

Restates the selection and weight rules of reference src/sycl/TopKSigMoid.cpp:96-176, src/sycl/BiasedTopK.cpp:100-170 and
src/sycl/MoE_fused_gate.cpp:130-330 with plain fp32 torch and EXPLICIT tie rules (iterative arg-max, ties -> lower
index; group ties -> lower group), so that the GPU index paths can be compared for equality. Pinned by
tests/test_oracle_golden.py against the reference tests' own references (fused_topk_sigmoid_torch_native,
biased_topk_torch_native, biased_grouped_topk_native; imported in tests/golden/make_golden.py)."""
import torch


def _pick(choice, k):
    """iterative arg-max, ties -> lower index. choice [T, E] fp32 -> ids [T, k] int64"""
    c = choice.clone()
    ids = []
    for _ in range(k):
        m = c.max(dim=-1, keepdim=True).values
        first = torch.argmax((c == m).to(torch.int8), dim=-1)  # first position of the maximum
        ids.append(first)
        c.scatter_(1, first.unsqueeze(1), float("-inf"))
    return torch.stack(ids, dim=1)


def _score(x, scoring):
    xf = x.float()
    if scoring == "sigmoid":
        return 1.0 / (1.0 + torch.exp(-xf))
    if scoring == "sqrtsoftplus":
        return torch.sqrt(torch.log1p(torch.exp(xf)))
    return torch.softmax(xf, dim=-1)


def topk_sigmoid(gating, topk, renormalize, correction_bias=None, routed_scaling_factor=1.0, num_fused_shared_experts=0):
    T, E = gating.shape
    score = _score(gating, "sigmoid")
    choice = score + correction_bias.float().unsqueeze(0) if correction_bias is not None else score
    routed = topk - num_fused_shared_experts
    ids = _pick(choice, routed)
    w = score.gather(1, ids)
    s = w.sum(dim=-1, keepdim=True)
    if renormalize:
        w = w * (routed_scaling_factor / (s + 1e-20))
    if num_fused_shared_experts:
        sid = torch.arange(E, E + num_fused_shared_experts).unsqueeze(0).expand(T, -1)
        sw = (torch.ones(T, 1) if renormalize else s / routed_scaling_factor).expand(T, num_fused_shared_experts)
        ids, w = torch.cat([ids, sid], 1), torch.cat([w, sw], 1)
    return w.float(), ids.to(torch.int32)


def biased_topk(x, bias, topk, scoring, num_fused_shared_experts=0, renormalize=False, routed_scaling_factor=1.0,
                apply_routed_scaling_factor_on_output=False):
    T, E = x.shape
    score = _score(x, scoring)
    routed = topk - num_fused_shared_experts
    ids = _pick(score + bias.float().unsqueeze(0), routed)
    w = score.gather(1, ids)
    s = w.sum(dim=-1, keepdim=True)
    if num_fused_shared_experts:
        sid = torch.arange(E, E + num_fused_shared_experts).unsqueeze(0).expand(T, -1)
        ids, w = torch.cat([ids, sid], 1), torch.cat([w, (s / routed_scaling_factor).expand(T, num_fused_shared_experts)], 1)
    norm = torch.where(s > 0, s, torch.ones_like(s)) if renormalize else torch.ones_like(s)
    w = (w / norm) * (routed_scaling_factor if apply_routed_scaling_factor_on_output else 1.0)
    return w.float(), ids.to(torch.int32)


def moe_fused_gate(x, bias, num_expert_group, topk_group, topk, num_fused_shared_experts=0, scoring="sigmoid", renormalize=True,
                   routed_scaling_factor=1.0, apply_routed_scaling_factor_on_output=False):
    T, E = x.shape
    G, gs = num_expert_group, E // num_expert_group
    score = _score(x, scoring)
    choice = score + bias.float().unsqueeze(0) if bias is not None else score
    top2 = choice.view(T, G, gs).topk(1 if scoring == "softmax" else 2, dim=-1).values.sum(dim=-1)  # [T, G]
    keep_groups = _pick(top2, topk_group)  # ties -> lower group
    gmask = torch.zeros(T, G, dtype=torch.bool).scatter_(1, keep_groups, True)
    masked = choice.masked_fill(~gmask.unsqueeze(-1).expand(T, G, gs).reshape(T, E), float("-inf"))
    routed = topk - num_fused_shared_experts
    ids = _pick(masked, routed)
    w = score.gather(1, ids)
    s = w.sum(dim=-1, keepdim=True)
    if num_fused_shared_experts:
        sid = torch.arange(E, E + num_fused_shared_experts).unsqueeze(0).expand(T, -1)
        ids, w = torch.cat([ids, sid], 1), torch.cat([w, (s / routed_scaling_factor).expand(T, num_fused_shared_experts)], 1)
    if renormalize:
        w = w * torch.where(s > 0, 1.0 / s, torch.zeros_like(s))
        if apply_routed_scaling_factor_on_output:
            w = w * routed_scaling_factor
    return w.float(), ids.to(torch.int32)
