"""GPU parity: MoE routing (exact integer paths), data movement, W4A16 grouped GEMM and fused_experts
vs the CPU oracle and the reference-generated golden vectors. Grids follow reference tests/test_topk_softmax.py:61-73,
tests/test_moe_align.py:141-155 and tests/test_moe_gemm.py:347-471."""
import numpy as np
import pytest
import torch
from conftest import load_golden

from oracle import activation as oact
from oracle import moe as omoe

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ topk_softmax
def check_topk(p, ref_idx, our_idx):
    """tests/test_topk_softmax.py:12-37: sets must match except where the swapped members have equal scores."""
    for r in (ref_idx != our_idx).any(dim=1).nonzero().flatten().tolist():
        a, b = set(our_idx[r].tolist()), set(ref_idx[r].tolist())
        assert sorted(p[r, list(a - b)].tolist()) == sorted(p[r, list(b - a)].tolist()), f"row {r}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_token", [2, 32, 4096])
@pytest.mark.parametrize("n_expert", [8, 32, 60, 256])
@pytest.mark.parametrize("n_topk", [1, 2, 4, 8])
@pytest.mark.parametrize("renormalize", [False, True])
def test_topk_softmax(sglk, dev, dtype, n_token, n_expert, n_topk, renormalize):
    g = torch.Generator().manual_seed(1024 + n_token + n_expert)
    for scale in (1.0, 2.0 * n_token):  # unit-scale logits, and the reference test's (mostly one-hot) scale
        gating = (torch.randn(n_token, n_expert, generator=g) * scale).to(dtype)
        w = torch.empty(n_token, n_topk, dtype=torch.float32, device=dev)
        idx = torch.empty(n_token, n_topk, dtype=torch.int32, device=dev)
        sglk.topk_softmax(w, idx, gating.to(dev), renormalize)
        rw, ridx, p = omoe.topk_softmax(gating, n_topk, renormalize)
        # the index path is exact: arg-max on the logits, lower index on ties
        assert torch.equal(idx.cpu(), ridx)
        # and it is a valid top-k of the softmax in the reference test's sense
        check_topk(p, torch.topk(p, n_topk, dim=-1).indices.to(torch.int32), idx.cpu())
        torch.testing.assert_close(w.cpu(), rw, rtol=1e-4, atol=1e-7)


def test_topk_softmax_ties_pick_lower_index(sglk, dev):
    gating = torch.zeros(3, 16, dtype=torch.bfloat16)
    gating[1, 5] = 1.0
    gating[2, 9] = gating[2, 3] = 2.0
    w = torch.empty(3, 3, dtype=torch.float32, device=dev)
    idx = torch.empty(3, 3, dtype=torch.int32, device=dev)
    sglk.topk_softmax(w, idx, gating.to(dev), False)
    assert idx.cpu().tolist() == [[0, 1, 2], [5, 0, 1], [3, 9, 0]]


# ---------------------------------------------------------------------------------- moe_align_block_size
@pytest.mark.parametrize("block_size", [32, 64, 128, 256])
@pytest.mark.parametrize("num_tokens,topk", [(1, 1), (2, 8), (16, 64), (128, 4), (1024, 2), (4096, 8), (4096, 64)])
@pytest.mark.parametrize("num_experts", [16, 64, 160, 256, 257, 264])
@pytest.mark.parametrize("pad", [True, False])
def test_moe_align_block_size(sglk, dev, block_size, num_tokens, topk, num_experts, pad):
    if topk > num_experts:
        pytest.skip("topk > experts")
    g = torch.Generator().manual_seed(num_tokens * 7 + topk + num_experts)
    ids = torch.argsort(torch.rand(num_tokens, num_experts, generator=g), dim=1)[:, :topk].contiguous()
    if num_tokens >= 16:
        ids[::5, 0] = -1  # tokens routed off-rank land in bucket 0 (tests/test_moe_align.py:183-193 passes E + 1)
    numel = ids.numel()
    max_pad = numel + (num_experts + 1) * (block_size - 1)
    sorted_ids = torch.full((max_pad,), numel, dtype=torch.int32, device=dev)
    if pad:
        sorted_ids.fill_(-7)
    expert_ids = torch.zeros(max_pad // block_size, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    cumsum = torch.empty(num_experts + 2, dtype=torch.int32, device=dev)
    sglk.moe_align_block_size(ids.to(dev), num_experts + 1, block_size, sorted_ids, expert_ids, total, cumsum, pad)
    rs, re, rtotal, rprefix = omoe.moe_align_block_size(ids.numpy(), num_experts + 1, block_size)
    assert total.item() == rtotal
    nblk = rtotal // block_size
    assert np.array_equal(expert_ids.cpu().numpy()[:nblk], re)
    got = sorted_ids.cpu().numpy()[:rtotal]
    # every bucket segment holds exactly the oracle's members (order inside a bucket is unspecified) + padding
    for b in range(num_experts + 1):
        lo, hi = rprefix[b], rprefix[b + 1]
        assert np.array_equal(np.sort(got[lo:hi]), np.sort(rs[lo:hi])), f"bucket {b}"
    counts = np.bincount(ids.numpy().reshape(-1) + 1, minlength=num_experts + 1)
    assert np.array_equal(cumsum.cpu().numpy()[: num_experts + 1], rprefix[: num_experts + 1] + counts)


# ------------------------------------------------------------------ prepare_moe_input / scatter / combine
@pytest.mark.parametrize("tokens,topk,E", [(1, 2, 8), (5, 2, 8), (300, 8, 64), (4096, 2, 8), (1000, 6, 256)])
@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
def test_prepare_scatter_combine(sglk, dev, tokens, topk, E, idt):
    g = torch.Generator().manual_seed(tokens + topk)
    ids = torch.argsort(torch.rand(tokens, E, generator=g), dim=1)[:, :topk].to(idt).contiguous()
    counts = torch.empty(E, dtype=idt, device=dev)
    ps1 = torch.empty(E, 3, dtype=idt, device=dev)
    ps2 = torch.empty(E, 3, dtype=idt, device=dev)
    a_map = torch.empty(tokens * topk, dtype=idt, device=dev)
    c_map = torch.empty(tokens * topk, dtype=idt, device=dev)
    sglk.prepare_moe_input(ids.to(dev), counts, ps1, ps2, a_map, c_map, E, 1024, 77)
    rc, r1, r2, ra, rcm = omoe.prepare_moe_input(ids.numpy(), E, 1024, 77)
    assert np.array_equal(counts.cpu().numpy(), rc)
    assert np.array_equal(ps1.cpu().numpy(), r1) and np.array_equal(ps2.cpu().numpy(), r2)
    assert np.array_equal(a_map.cpu().numpy(), ra) and np.array_equal(c_map.cpu().numpy(), rcm)
    if idt != torch.int32:
        return
    hidden = 520  # not a multiple of the 256-thread vector stride
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(tokens, hidden, generator=g).to(dt)
        out = torch.zeros(tokens * topk, hidden, dtype=dt, device=dev)
        sglk.scatter_tokens_to_experts(x.to(dev), c_map, out)
        assert torch.equal(out.cpu(), x[torch.from_numpy(ra).long()])
        w = torch.rand(tokens, topk, generator=g)
        y = torch.empty(tokens, hidden, dtype=dt, device=dev)
        for rsf in (None, 2.5):
            sglk.apply_shuffle_mul_sum(out, y, c_map, w.to(dev), rsf)
            t = x.float().unsqueeze(1) * w.unsqueeze(-1)
            if rsf:
                t = t * rsf
            acc = torch.zeros(tokens, hidden)
            for j in range(topk):
                acc = acc + t[:, j]
            assert torch.equal(y.cpu(), acc.to(dt)), "fp32 slot-order accumulation must be bit-exact"
        sglk.apply_shuffle_mul_sum(out, y, c_map, None)
        torch.testing.assert_close(y.cpu().float(), x.float() * topk, rtol=1e-2, atol=1e-2)


# ------------------------------------------------------------------------------- W4A16 grouped GEMM
def make_int4(E, N, K, gs, dtype, explicit_zero, g):
    if explicit_zero:
        codes = torch.randint(0, 16, (E, N, K), generator=g, dtype=torch.int16)
        zeros = torch.randint(3, 13, (E, N, K // gs), generator=g).to(dtype)
    else:
        codes = torch.randint(-8, 8, (E, N, K), generator=g, dtype=torch.int16)
        zeros = None
    scales = (torch.rand(E, N, K // gs, generator=g) * 0.02 + 0.005).to(dtype)
    nib = codes & 0xF
    packed = (nib[..., 0::2] | (nib[..., 1::2] << 4)).to(torch.uint8)
    return packed, scales, zeros


@pytest.mark.parametrize("explicit_zero", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gs", [32, 64, 128, 256])
@pytest.mark.parametrize("rows,N,K", [([2] * 8, 128, 256), ([0, 5, 17, 0, 1, 33, 0, 129], 200, 512),
                                      ([300, 0, 40, 7], 1024, 1280), ([1] * 8, 4096, 1024),
                                      # few rows, long K (groups of 128: the four waves of a workgroup split K)
                                      ([3, 0, 16, 7], 104, 8192), ([1] * 8, 256, 10240),
                                      # many rows: the eight-wave 64 x 256 tile
                                      ([130, 200, 112, 150], 512, 1024),
                                      # prefill row counts: the dense tile pipeline of moe_persist.hip (groups of 128, no zero points)
                                      ([300, 200, 513, 256], 512, 1280), ([700, 0, 1, 900], 200, 256)])
def test_moe_grouped_mm_w4a16(sglk, dev, explicit_zero, dtype, gs, rows, N, K):
    if K % gs:
        pytest.skip("K not a multiple of the group")
    g = torch.Generator().manual_seed(len(rows) * 31 + N + K + gs)
    E = len(rows)
    total = sum(rows)
    act = (torch.randn(total, K, generator=g) * 0.1).to(dtype)
    packed, scales, zeros = make_int4(E, N, K, gs, dtype, explicit_zero, g)
    bias = torch.randn(E, N, generator=g) * 0.01 if (N % 256 == 0) else None
    rows_t = torch.tensor(rows, dtype=torch.int32)
    out = torch.full((total, N), float("nan"), dtype=dtype, device=dev)
    torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(
        out, act.to(dev), packed.view(torch.int8 if explicit_zero else torch.uint8).to(dev), scales.to(dev),
        zeros.to(dev) if zeros is not None else None, bias.to(dev) if bias is not None else None, rows_t.to(dev), E,
        True, gs)
    ref = omoe.moe_grouped_mm_w4a16(act, packed, scales, zeros, bias, rows_t, gs)
    torch.testing.assert_close(out.cpu(), ref, rtol=5e-2, atol=2e-2)  # reference tolerance (tests/test_moe_gemm.py:386)
    # the kernel keeps the codes exact and scales in fp32; against that (more accurate) definition it is tight
    codes = omoe.unpack_int4(packed, signed=zeros is None).float()
    z = zeros.float().repeat_interleave(gs, dim=-1) if zeros is not None else 0.0
    w_exact = (codes - z) * scales.float().repeat_interleave(gs, dim=-1)
    r0, exact = 0, torch.empty(total, N)
    for e, r in enumerate(rows):
        exact[r0:r0 + r] = act[r0:r0 + r].float() @ w_exact[e].t() + (bias[e] if bias is not None else 0.0)
        r0 += r
    # (prefill row counts run on the tile pipeline of moe_persist.hip, which rounds (code - zero) * scale once to the activation
    # type as the reference's dequantisation does, gemm_xe2.hpp:52-76; an expert's last rows stay on the streaming kernels:
    # every element is tight against one of the two definitions - the oracle's for the tile pipeline's rows)
    o = out.cpu().float()
    near_exact = torch.isclose(o, exact.to(dtype).float(), rtol=1e-2, atol=2e-3)
    if sum(rows) < 88 * E:
        assert near_exact.all(), "the streaming kernels keep the codes exact"
    else:
        assert (near_exact | torch.isclose(o, ref.float(), rtol=1e-2, atol=2e-3)).all()


@pytest.mark.parametrize("act_type", [1, 2, 3, 4, 5])  # silu, gelu (tanh), relu2, DeepSeek-V4 clamped swiglu, gpt-oss swiglu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gs,explicit_zero", [(128, False), (32, True), (64, False)])
@pytest.mark.parametrize("rows,N,K", [([2] * 8, 256, 256), ([0, 5, 17, 0, 1, 33, 0, 129], 416, 512),
                                      ([300, 0, 40, 7], 1024, 1280), ([1] * 8, 4096, 1024),
                                      ([130, 200, 112, 150], 512, 1024), ([3, 0, 16, 7], 208, 8192),
                                      ([300, 200, 513, 256], 1024, 1280), ([260, 250, 300, 310], 384, 256)])  # (moe_persist.hip)
def test_moe_grouped_mm_w4a16_fused_act(sglk, dev, act_type, dtype, gs, explicit_zero, rows, N, K):
    """the gate / up activation (or relu2) in the epilogue of the W4A16 GEMM (authored op moe_grouped_mm_nt_w4a16_act):
    act(gate) * up from the fp32 accumulators, one rounding, against the exact-code fp32 definition; and within the
    reference tolerance of the two-launch route (GEMM rounded to T, then silu_and_mul / gelu_tanh_and_mul on it)."""
    g = torch.Generator().manual_seed(len(rows) * 31 + N + K + gs + act_type)
    E, total = len(rows), sum(rows)
    act = (torch.randn(total, K, generator=g) * 0.1).to(dtype)
    packed, scales, zeros = make_int4(E, N, K, gs, dtype, explicit_zero, g)
    bias = torch.randn(E, N, generator=g) * 0.05 if (N % 256 == 0) else None
    rows_t = torch.tensor(rows, dtype=torch.int32)
    gated = act_type != 3
    out = torch.full((total, N // 2 if gated else N), float("nan"), dtype=dtype, device=dev)
    wq = packed.view(torch.int8 if explicit_zero else torch.uint8).to(dev)
    d = lambda t: t.to(dev) if t is not None else None
    limit = 0.25  # (small enough that the clamp of act_type 4 bites on these inputs)
    alpha = 1.702
    torch.ops.sgl_kernel.moe_grouped_mm_nt_w4a16_act(out, act.to(dev), wq, scales.to(dev), d(zeros), d(bias), rows_t.to(dev),
                                                     E, True, gs, act_type, limit, None, alpha)
    codes = omoe.unpack_int4(packed, signed=zeros is None).float()
    z = zeros.float().repeat_interleave(gs, dim=-1) if zeros is not None else 0.0
    w_exact = (codes - z) * scales.float().repeat_interleave(gs, dim=-1)
    r0, x = 0, torch.empty(total, N)
    for e, r in enumerate(rows):
        x[r0:r0 + r] = act[r0:r0 + r].float() @ w_exact[e].t() + (bias[e] if bias is not None else 0.0)
        r0 += r
    if act_type == 1:
        ref = torch.nn.functional.silu(x[:, :N // 2]) * x[:, N // 2:]
    elif act_type == 2:
        ref = torch.nn.functional.gelu(x[:, :N // 2], approximate="tanh") * x[:, N // 2:]
    elif act_type == 4:  # reference silu_and_mul_clamp: gate = min(gate, limit), up = clamp(up, +-limit)
        ref = torch.nn.functional.silu(x[:, :N // 2].clamp(max=limit)) * x[:, N // 2:].clamp(-limit, limit)
        assert (x[:, :N // 2] > limit).any() and (x[:, N // 2:].abs() > limit).any()
    elif act_type == 5:  # gate = weight rows 0, 2, .., up = rows 1, 3, .. (reference moe_kernel.hpp:109-125, activation.hpp:35-41)
        gt, up = x[:, 0::2].clamp(max=limit), x[:, 1::2].clamp(-limit, limit)
        ref = gt * torch.sigmoid(alpha * gt) * (up + 1.0)
        assert (x[:, 0::2] > limit).any() and (x[:, 1::2].abs() > limit).any()
    else:
        ref = torch.relu(x) ** 2
    torch.testing.assert_close(out.cpu().float(), ref.to(dtype).float(), rtol=1e-2, atol=2e-3)
    # the two-launch route of the reference (its own tolerance, tests/test_moe_gemm.py:386)
    gu = torch.empty(total, N, dtype=dtype, device=dev)
    torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(gu, act.to(dev), wq, scales.to(dev), d(zeros), d(bias), rows_t.to(dev), E,
                                                      True, gs)
    if act_type == 4:
        guf = gu.float()
        two = torch.nn.functional.silu(guf[:, :N // 2].clamp(max=limit)) * guf[:, N // 2:].clamp(-limit, limit)
    elif act_type == 5:  # (the reference's W4A16 route: swiglu_gpt_oss_sigmoid_alpha on the rounded [rows, 2I] product)
        two = torch.ops.sgl_kernel.swiglu_gpt_oss_sigmoid_alpha(gu, alpha, limit)
    elif gated:
        two = torch.empty(total, N // 2, dtype=dtype, device=dev)
        (torch.ops.sgl_kernel.silu_and_mul if act_type == 1 else torch.ops.sgl_kernel.gelu_tanh_and_mul)(two, gu)
    else:
        two = torch.relu(gu.float()) ** 2
    torch.testing.assert_close(out.cpu().float(), two.cpu().float(), rtol=5e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act_type", [1, 3, 4, 5])
@pytest.mark.parametrize("fmt", ["int4", "int4_zp", "mxfp4"])
@pytest.mark.parametrize("rows,N,K,tokens", [([2] * 8, 256, 256, 3), ([0, 5, 17, 0, 1, 33, 0, 70], 416, 512, 40),
                                             ([1] * 8, 4096, 1024, 1), ([3, 0, 16, 7], 208, 4096, 9),
                                             ([130, 200, 112, 150], 512, 1024, 300)])
def test_moe_grouped_mm_w4a16_row_map_is_the_gather(sglk, dev, dtype, act_type, fmt, rows, N, K, tokens):
    """row_map of the authored op (the token gather of fused_experts, reference shuffle_rows, python/sgl_kernel/moe.py:739,
    folded into GEMM 1's staging loads): the mapped call on the tokens equals, bit for bit, the plain call on the rows
    gathered beforehand - at every size, also where the unmapped call would run on the tile pipeline (the last case: that
    one is compared within the reference tolerance, the two pipelines round differently)."""
    g = torch.Generator().manual_seed(sum(rows) + N + K + act_type)
    E, total = len(rows), sum(rows)
    x = (torch.randn(tokens, K, generator=g) * 0.1).to(dtype).to(dev)
    row_map = torch.randint(0, tokens, (total,), generator=g, dtype=torch.int32).to(dev)
    if fmt == "mxfp4":
        packed = torch.randint(0, 256, (E, N, K // 2), generator=g, dtype=torch.uint8)
        scales, zeros, gs, is_int4 = torch.randint(118, 126, (E, N, K // 32), generator=g, dtype=torch.uint8), None, 32, False
    else:
        gs, is_int4 = 128, True
        packed, scales, zeros = make_int4(E, N, K, gs, dtype, fmt == "int4_zp", g)
        packed = packed.view(torch.int8 if fmt == "int4_zp" else torch.uint8)
    bias = (torch.randn(E, N, generator=g) * 0.05).to(dev) if N % 256 == 0 else None
    d = lambda t: t.to(dev) if t is not None else None
    rows_t = torch.tensor(rows, dtype=torch.int32, device=dev)
    cols = N if act_type == 3 else N // 2
    mapped = torch.full((total, cols), float("nan"), dtype=dtype, device=dev)
    plain = torch.full((total, cols), float("nan"), dtype=dtype, device=dev)
    op = torch.ops.sgl_kernel.moe_grouped_mm_nt_w4a16_act
    op(mapped, x, packed.to(dev), scales.to(dev), d(zeros), bias, rows_t, E, is_int4, gs, act_type, 0.25, row_map, 1.702)
    op(plain, x[row_map.long()].contiguous(), packed.to(dev), scales.to(dev), d(zeros), bias, rows_t, E, is_int4, gs, act_type, 0.25,
       None, 1.702)
    assert torch.isfinite(mapped.float()).all()
    if total < 88 * E:
        assert torch.equal(mapped, plain)
    else:
        torch.testing.assert_close(mapped.float(), plain.float(), rtol=5e-2, atol=2e-2)
    with pytest.raises(RuntimeError, match="row_map"):
        op(mapped, x, packed.to(dev), scales.to(dev), d(zeros), bias, rows_t, E, is_int4, gs, act_type, 0.25, row_map.long())
    with pytest.raises(RuntimeError, match="row_map's length"):
        op(mapped, x, packed.to(dev), scales.to(dev), d(zeros), bias, rows_t, E, is_int4, gs, act_type, 0.25, row_map[:-1])


def _tail_rows_mask(rows, block=128):
    """rows of an expert's remainder of 1 .. block / 2 rows behind its last full row block (the streaming kernels' share)"""
    m = []
    for r in rows:
        rem = r % block
        t = rem if 1 <= rem <= block // 2 else 0
        m += [False] * (r - t) + [True] * t
    return torch.tensor(m, dtype=torch.bool)


@pytest.mark.parametrize("fmt,dtype", [("int4", torch.bfloat16), ("int4", torch.float16), ("int4_zp", torch.bfloat16),
                                       ("int4_zp", torch.float16), ("mxfp4", torch.bfloat16)])
@pytest.mark.parametrize("rows,N,K,gs", [([128] * 8, 512, 1024, 128),
                                         ([124, 106, 126, 146, 137, 133, 124, 128], 512, 1024, 128),
                                         ([130, 200, 112, 150], 200, 1280, 128),   # edge column block, 72-row remainder block
                                         ([191, 96, 129, 64 + 128], 256, 2048, 64),
                                         ([97] * 16, 256, 512, 32),                # K / 2 = four K blocks: the shortest unit
                                         ([130, 200, 112, 150], 512, 1280, 256),   # K / 2 is not whole groups: no split
                                         # from an average of 192 rows per expert: 256-row blocks, remainders of up to 128 rows
                                         ([256] * 8, 512, 1024, 128), ([300, 200, 513, 256], 512, 1280, 128),
                                         ([236, 276, 250, 291, 232, 251, 269, 243], 200, 1024, 64)])
def test_moe_grouped_mm_w4a16_splitk(sglk, dev, fmt, dtype, rows, N, K, gs):
    """The down projection's K split (moe_persist.hip, KSPL = 2): full 128-row blocks as two fp32 partial sums in ws, remainders
    of 1 .. 64 rows in out; bf16(ws[0] + ws[1]) against the oracle at the reference's tolerance and against the unsplit op."""
    g = torch.Generator().manual_seed(N + K + gs + len(rows))
    E, total = len(rows), sum(rows)
    act = (torch.randn(total, K, generator=g) * 0.1).to(dtype)
    if fmt == "mxfp4":
        if gs != 128:
            pytest.skip("mxfp4 has one group size")
        gs = 32
        packed = torch.randint(0, 256, (E, N, K // 2), generator=g, dtype=torch.uint8)
        scales, zeros, is_int4 = torch.randint(118, 126, (E, N, K // 32), generator=g, dtype=torch.uint8), None, False
    else:
        packed, scales, zeros = make_int4(E, N, K, gs, dtype, fmt == "int4_zp", g)
        is_int4 = True
    d = lambda t: t.to(dev) if t is not None else None
    rows_t = torch.tensor(rows, dtype=torch.int32, device=dev)
    op = torch.ops.sgl_kernel
    applies = op.moe_w4a16_splitk_applies(total, E, N, K, gs, is_int4, dtype == torch.bfloat16)
    assert applies == (0 if (K // 2) % gs else 128 if total < 152 * E else 256)
    y = torch.full((total, N), float("nan"), dtype=dtype, device=dev)
    ws = torch.full((2, total, N), float("nan"), dtype=torch.float32, device=dev)
    used = op.moe_grouped_mm_nt_w4a16_splitk(y, ws, act.to(dev), packed.to(dev), scales.to(dev), d(zeros), rows_t, E, is_int4, gs)
    assert used == applies
    plain = torch.full((total, N), float("nan"), dtype=dtype, device=dev)
    op.moe_grouped_mm_nt_xe20_w4a16(plain, act.to(dev), packed.to(dev), scales.to(dev), d(zeros), None, rows_t, E, is_int4, gs)
    if not used:
        assert torch.equal(y, plain), "without the split the call is the plain op"
        assert torch.isnan(ws).all(), "... and does not touch ws"
        return
    tail = _tail_rows_mask(rows, used)
    assert torch.isnan(y.cpu()[~tail].float()).all(), "rows of the split blocks are not written to out"
    assert torch.isnan(ws.cpu()[:, tail]).all(), "remainder rows are not written to ws"
    got = torch.where(tail[:, None], y.cpu().float(), (ws[0] + ws[1]).to(dtype).cpu().float())
    assert torch.isfinite(got).all()
    if fmt == "mxfp4":
        ref = omoe.moe_grouped_mm_w4a16(act, packed, scales, None, None, rows_t.cpu(), 32, mxfp4=True)
    else:
        ref = omoe.moe_grouped_mm_w4a16(act, packed, scales, zeros, None, rows_t.cpu(), gs)
    torch.testing.assert_close(got.to(dtype), ref, rtol=5e-2, atol=2e-2)  # reference tolerance (tests/test_moe_gemm.py:386)
    # rows of the split blocks: the same tile arithmetic in another summation order - tight. (The remainders run on the streaming
    # kernels, which keep the codes exact and scale in fp32, while the unsplit call may take them into the tile pipeline, which
    # rounds code * scale once as the reference does - INTEGRATION.md note 16: those rows have the oracle check above.)
    torch.testing.assert_close(got[~tail], plain.cpu().float()[~tail], rtol=1e-2, atol=2e-3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block", [128, 256])
@pytest.mark.parametrize("rows,topk,hidden", [([128] * 8, 2, 512), ([124, 106, 126, 146, 137, 133, 124, 128], 2, 4096),
                                              ([130, 200, 112, 150, 0, 64, 65, 1], 6, 136), ([300, 129, 513, 256, 128], 2, 256)])
def test_apply_shuffle_mul_sum_splitk(sglk, dev, dt, block, rows, topk, hidden):
    """out[t] = T(sum_j T(x[perm[t, j]]) * w[t, j] * rsf), x[r] = y[r] for an expert's remainder of 1 .. 64 rows and
    T(ws[0][r] + ws[1][r]) otherwise: bit-exact against that definition."""
    g = torch.Generator().manual_seed(sum(rows) + hidden)
    total = sum(rows)
    total -= total % topk
    rows = list(rows)
    rows[0] -= sum(rows) - total
    tokens = total // topk
    y = torch.randn(total, hidden, generator=g).to(dt)
    ws = torch.randn(2, total, hidden, generator=g)
    perm = torch.randperm(total, generator=g).to(torch.int32).view(tokens, topk)
    w = torch.rand(tokens, topk, generator=g)
    out = torch.empty(tokens, hidden, dtype=dt, device=dev)
    torch.ops.sgl_kernel.apply_shuffle_mul_sum_splitk(y.to(dev), ws.to(dev), out, perm.to(dev),
                                                      torch.tensor(rows, dtype=torch.int32, device=dev), block, 2.5, w.to(dev))
    tail = _tail_rows_mask(rows, block)
    x = torch.where(tail[:, None], y.float(), (ws[0] + ws[1]).to(dt).float())
    acc = torch.zeros(tokens, hidden)
    for j in range(topk):
        acc = acc + (x[perm[:, j].long()] * w[:, j:j + 1]) * 2.5
    assert torch.equal(out.cpu(), acc.to(dt))


@pytest.mark.parametrize("fmt", ["int4", "int4_zp", "mxfp4"])
def test_moe_grouped_mm_w4a16_row_map_without_activation_on_the_k_split_shape(sglk, dev, fmt):
    """fused_act = 0 with a row_map at the shape the K-split kernel takes (few rows per expert, K = 8192, groups of 128): that
    kernel reads expert-contiguous rows, so a mapped call must not be routed to it (round 4 did: wrong rows, reads past the
    [src_rows, K] activations). The mapped call equals the plain call on the gathered rows bit for bit."""
    dtype, rows, N, K, tokens = torch.bfloat16, [3, 0, 5, 1, 2, 0, 4, 1], 256, 8192, 6
    g = torch.Generator().manual_seed(77)
    E, total = len(rows), sum(rows)
    x = (torch.randn(tokens, K, generator=g) * 0.1).to(dtype).to(dev)
    row_map = torch.randint(0, tokens, (total,), generator=g, dtype=torch.int32).to(dev)
    if fmt == "mxfp4":
        packed = torch.randint(0, 256, (E, N, K // 2), generator=g, dtype=torch.uint8)
        scales, zeros, gs, is_int4 = torch.randint(118, 126, (E, N, K // 32), generator=g, dtype=torch.uint8), None, 32, False
    else:
        gs, is_int4 = 128, True
        packed, scales, zeros = make_int4(E, N, K, gs, dtype, fmt == "int4_zp", g)
        packed = packed.view(torch.int8 if fmt == "int4_zp" else torch.uint8)
    d = lambda t: t.to(dev) if t is not None else None
    rows_t = torch.tensor(rows, dtype=torch.int32, device=dev)
    mapped = torch.full((total, N), float("nan"), dtype=dtype, device=dev)
    plain = torch.full((total, N), float("nan"), dtype=dtype, device=dev)
    op = torch.ops.sgl_kernel.moe_grouped_mm_nt_w4a16_act
    op(mapped, x, packed.to(dev), scales.to(dev), d(zeros), None, rows_t, E, is_int4, gs, 0, 0.0, row_map)
    gathered = x[row_map.long()].contiguous()
    op(plain, gathered, packed.to(dev), scales.to(dev), d(zeros), None, rows_t, E, is_int4, gs, 0, 0.0)
    assert torch.isfinite(mapped.float()).all()
    # (the plain call may run on the K-split kernel, whose four waves add their K quarters in another order: tolerance there)
    torch.testing.assert_close(mapped.float(), plain.float(), rtol=2e-2, atol=2e-2)


def test_fused_experts_swiglu_limit(sglk, dev):
    """DeepSeek-V4 clamp (reference moe.py:699-709, tests/test_fused_experts_mxfp4_dsv4_shapes.py:59-61): with a limit no
    pre-activation reaches, the clamped layer equals the plain silu layer bit for bit; with inputs scaled up it equals the
    layer built from the same ops with the clamp done by hand."""
    g = torch.Generator().manual_seed(21)
    T, E, topk, H, I, gs, dt = 48, 8, 2, 512, 1024, 128, torch.bfloat16
    x = (torch.randn(T, H, generator=g) * 0.1).to(dt).to(dev)
    w1, s1, _ = make_int4(E, 2 * I, H, gs, dt, False, g)
    w2, s2, _ = make_int4(E, H, I, gs, dt, False, g)
    w1, s1, w2, s2 = w1.view(torch.uint8).to(dev), s1.to(dev), w2.view(torch.uint8).to(dev), s2.to(dev)
    tw = torch.rand(T, topk, generator=g).to(dev)
    ti = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(T)]).to(torch.int32).to(dev)
    kw = dict(use_int4_w4a16=True, w1_scale=s1, w2_scale=s2)
    plain = sglk.fused_experts(x, w1, w2, tw, ti, **kw)
    clamped = sglk.fused_experts(x, w1, w2, tw, ti, swiglu_limit=10, **kw)
    assert torch.equal(plain, clamped)  # (|pre-activations| stay far below 10 here)
    big = sglk.fused_experts(x * 64, w1, w2, tw, ti, swiglu_limit=10, **kw)
    big_plain = sglk.fused_experts(x * 64, w1, w2, tw, ti, **kw)
    assert not torch.equal(big, big_plain)
    # the same layer by hand: one expert-contiguous pass through the unfused GEMM, clamp, silu * up, second GEMM, combine
    ref = torch.zeros(T, H, dtype=torch.float32, device=dev)
    for e in range(E):
        rows = (ti == e).any(dim=1).nonzero().flatten()
        if rows.numel() == 0:
            continue
        cnt = torch.zeros(E, dtype=torch.int32, device=dev)
        cnt[e] = rows.numel()
        gu = torch.empty(rows.numel(), 2 * I, dtype=dt, device=dev)
        torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(gu, (x * 64)[rows].contiguous(), w1, s1, None, None, cnt, E, True, gs)
        guf = gu.float()
        h = (torch.nn.functional.silu(guf[:, :I].clamp(max=10.0)) * guf[:, I:].clamp(-10.0, 10.0)).to(dt)
        y = torch.empty(rows.numel(), H, dtype=dt, device=dev)
        torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(y, h, w2, s2, None, None, cnt, E, True, gs)
        wsel = (tw * (ti == e)).sum(dim=1)[rows]
        ref[rows] += y.float() * wsel[:, None]
    torch.testing.assert_close(big.float(), ref, rtol=5e-2, atol=5e-2 * ref.abs().max().item())


def test_w4a16_golden(sglk, dev):
    for c in load_golden("moe_w4a16")["grouped_mm"]:
        E = c["packed"].shape[0]
        rows = torch.full((E,), c["rows_per_expert"], dtype=torch.int32, device=dev)
        out = torch.empty_like(c["out"], device=dev)
        torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(
            out, c["act"].to(dev), c["packed"].to(dev), c["scales"].to(dev),
            c["zeros"].to(dev) if c["zeros"] is not None else None, None, rows, E, True, c["group_size"])
        torch.testing.assert_close(out.cpu(), c["out"], rtol=5e-2, atol=2e-2)


def test_fused_experts_golden(sglk, dev):
    d = lambda t: t.to(dev) if t is not None else None
    for c in load_golden("moe_w4a16")["fused"]:
        out = sglk.fused_experts(d(c["x"]), d(c["w1"].view(torch.int8)), d(c["w2"].view(torch.int8)),
                                 d(c["topk_weights"]), d(c["topk_ids"]), d(c["b1"]), d(c["b2"]),
                                 activation=c["activation"], use_int4_w4a16=True, w1_scale=d(c["w1_scale"]),
                                 w2_scale=d(c["w2_scale"]), w1_zp=d(c["w1_zp"]), w2_zp=d(c["w2_zp"]))
        torch.testing.assert_close(out.cpu(), c["out"], rtol=1e-1, atol=2e-2)  # tests/test_moe_gemm.py:471


# ------------------------------------------------------------------------------- mxfp4 weights (e2m1 + E8M0 / 32)
def make_mxfp4(E, N, K, g, exp_lo=110, exp_hi=135):
    """Random e2m1 nibbles and E8M0 scale bytes: every bit pattern is a valid weight, so no quantiser is needed."""
    packed = torch.randint(0, 256, (E, N, K // 2), generator=g, dtype=torch.int16).to(torch.uint8)
    scales = torch.randint(exp_lo, exp_hi, (E, N, K // 32), generator=g, dtype=torch.int16).to(torch.uint8)
    return packed, scales


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,N,K", [([2] * 8, 1024, 1024), ([6] * 8, 1024, 1024), ([33] * 8, 1024, 1024),
                                      ([129] * 8, 1024, 1024), ([0, 5, 17, 0, 1, 33, 0, 129], 200, 512),
                                      ([300, 0, 40, 7], 1024, 1280), ([1] * 8, 4096, 1024),
                                      # few rows, long K: the four waves of a workgroup split K (bf16: conversion path)
                                      ([3, 0, 16, 7], 104, 8192), ([1] * 8, 256, 10240),
                                      ([300, 200, 513, 256], 1024, 1280), ([700, 0, 1, 900], 200, 192)])  # (moe_persist.hip)
def test_moe_grouped_mm_w4a16_mxfp4(sglk, dev, dtype, rows, N, K):
    """reference tests/test_moe_gemm.py:805-885 (rows per expert {2, 6, 33, 129}, E=8, K=1024, N=2*512) + ragged rows"""
    g = torch.Generator().manual_seed(len(rows) * 17 + N + K)
    E, total = len(rows), sum(rows)
    act = (torch.randn(total, K, generator=g) * 0.1).to(dtype)
    packed, scales = make_mxfp4(E, N, K, g, 113, 124)  # |w| <= 6 * 2^-4: products stay far inside fp16
    bias = torch.randn(E, N, generator=g) * 0.01 if (N % 256 == 0) else None
    rows_t = torch.tensor(rows, dtype=torch.int32)
    out = torch.full((total, N), float("nan"), dtype=dtype, device=dev)
    torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(
        out, act.to(dev), packed.to(dev), scales.to(dev), None, bias.to(dev) if bias is not None else None,
        rows_t.to(dev), E, False, 32)
    ref = omoe.moe_grouped_mm_w4a16(act, packed, scales, None, bias, rows_t, 32, mxfp4=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-1, atol=1e-2)  # reference tolerance (test_moe_gemm.py:846)
    # the dequantised weights are exact in T, so only the fp32 summation order differs from the oracle
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=1e-2, atol=2e-3)


@pytest.mark.parametrize("wdt", [torch.int8, torch.uint8])
@pytest.mark.parametrize("sdt", ["uint8", "float8_e8m0fnu"])
def test_mxfp4_input_dtypes_and_scale_range(sglk, dev, wdt, sdt):
    """reference tests/test_moe_gemm.py:888-904: int8 / uint8 codes and uint8 / float8_e8m0fnu scales carry the same
    bits. The scale bytes here span 2^-127 (byte 0) .. 2^20 so the E8M0 decode is checked at its ends (bf16 only: the
    products leave fp16's range)."""
    if not hasattr(torch, sdt):
        pytest.skip(f"torch has no {sdt}")
    g = torch.Generator().manual_seed(5)
    E, N, K, rows = 4, 256, 512, [3, 0, 65, 9]
    act = (torch.randn(sum(rows), K, generator=g) * 0.1).to(torch.bfloat16)
    packed, scales = make_mxfp4(E, N, K, g, 0, 148)
    scales[0, 0, :4] = torch.tensor([0, 1, 2, 147], dtype=torch.uint8)
    rows_t = torch.tensor(rows, dtype=torch.int32)
    out = torch.empty(sum(rows), N, dtype=torch.bfloat16, device=dev)
    torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(
        out, act.to(dev), packed.view(wdt).to(dev), scales.to(dev).view(getattr(torch, sdt)), None, None, rows_t.to(dev),
        E, False, 32)
    ref = omoe.moe_grouped_mm_w4a16(act, packed, scales, None, None, rows_t, 32, mxfp4=True)
    o, r = out.cpu().float(), ref.float()
    assert torch.isfinite(o).all()
    # outputs span ~40 binades: bound the error by each row's largest term instead of an absolute floor
    assert ((o - r).abs() <= 1e-2 * r.abs() + 1e-3 * r.abs().amax(dim=1, keepdim=True)).all()


def test_mxfp4_golden(sglk, dev):
    g = load_golden("moe_w4a16")
    for c in g["mxfp4_grouped_mm"]:
        E = c["packed"].shape[0]
        rows = torch.full((E,), c["rows_per_expert"], dtype=torch.int32, device=dev)
        out = torch.empty_like(c["out"], device=dev)
        torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(out, c["act"].to(dev), c["packed"].to(dev), c["scales"].to(dev),
                                                          None, None, rows, E, False, 32)
        torch.testing.assert_close(out.cpu(), c["out"], rtol=1e-1, atol=1e-2)
    for c in g["mxfp4_fused"]:
        out = sglk.fused_experts(c["x"].to(dev), c["w1"].to(dev), c["w2"].to(dev), c["topk_weights"].to(dev),
                                 c["topk_ids"].to(dev), use_mxfp4_w4a16=True, w1_scale=c["w1_scale"].to(dev),
                                 w2_scale=c["w2_scale"].to(dev))
        torch.testing.assert_close(out.cpu(), c["out"], rtol=1e-1, atol=1e-2)  # tests/test_moe_gemm.py:580


@pytest.mark.parametrize("T,topk,E,H,I", [(1, 1, 8, 128, 128), (33, 2, 8, 1024, 512), (222, 6, 64, 128, 512),
                                          (33, 6, 64, 1024, 128)])
@pytest.mark.parametrize("bias", [False, True])
def test_fused_experts_mxfp4(sglk, dev, T, topk, E, H, I, bias):
    """reference tests/test_moe_gemm.py:555-640: fused_experts(use_mxfp4_w4a16=True) against the MLP on the
    dequantised weights (grid sampled from its [1,33,222] x [1,2,6] x [8,64] x [128,1024] x [128,512])"""
    g = torch.Generator().manual_seed(T + topk + E + H + I)
    dt = torch.bfloat16
    x = (torch.randn(T, H, generator=g) * 0.1).to(dt)
    w1, s1 = make_mxfp4(E, 2 * I, H, g, 119, 123)
    w2, s2 = make_mxfp4(E, H, I, g, 119, 123)
    b1 = torch.randn(E, 2 * I, generator=g) * 0.005 if bias else None
    b2 = torch.randn(E, H, generator=g) * 0.005 if bias else None
    score = torch.softmax(torch.randn(T, E, generator=g), dim=-1)
    tw, ids = torch.topk(score, topk)
    d = lambda t: t.to(dev) if t is not None else None
    out = sglk.fused_experts(d(x), d(w1.view(torch.int8)), d(w2), d(tw), d(ids), d(b1), d(b2), use_mxfp4_w4a16=True,
                             w1_scale=d(s1), w2_scale=d(s2))
    ref = omoe.fused_experts_int4(x, w1, w2, tw, ids, s1, s2, None, None, b1, b2, mxfp4=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-1, atol=1e-2)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=3e-2, atol=1e-2)


@pytest.mark.parametrize("T", [1, 32, 512])
def test_fused_experts_mixtral_shape_sampled(sglk, dev, T):
    """BASELINE configs[4] shape class (E=8, top-2, H=4096, int4 g=128) at a reduced intermediate size so the CPU
    oracle stays in seconds; routing through topk_softmax(renormalize=True) as SGLang does."""
    E, k, H, I, gs, dt = 8, 2, 4096, 1792, 128, torch.bfloat16
    g = torch.Generator().manual_seed(T)
    x = (torch.randn(T, H, generator=g) * 0.1).to(dt)
    w1, s1, _ = make_int4(E, 2 * I, H, gs, dt, False, g)
    w2, s2, _ = make_int4(E, H, I, gs, dt, False, g)
    logits = torch.randn(T, E, generator=g).to(dt)
    tw = torch.empty(T, k, dtype=torch.float32, device=dev)
    ids = torch.empty(T, k, dtype=torch.int32, device=dev)
    sglk.topk_softmax(tw, ids, logits.to(dev), True)
    out = sglk.fused_experts(x.to(dev), w1.view(torch.int8).to(dev), w2.view(torch.int8).to(dev), tw, ids,
                             use_int4_w4a16=True, w1_scale=s1.to(dev), w2_scale=s2.to(dev))
    ref = omoe.fused_experts_int4(x, w1, w2, tw.cpu(), ids.cpu().long(), s1, s2)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-1, atol=2e-2)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=3e-2, atol=1e-2)


# ---------------------------------------------------------------------- 16-bit weights
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows", [[2] * 8, [0, 17, 1, 0, 130, 3, 64, 33], [300] + [0] * 7,
                                  [200, 130, 97, 255, 128, 100, 190, 140],  # (the eight-wave 128 x 256 tile)
                                  [300, 200, 513, 256, 260, 250, 300, 310]])  # (moe_persist.hip where K % 64 == 0, no bias)
@pytest.mark.parametrize("N,K", [(128, 256), (352, 2816), (2816, 176), (1024, 1000)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_grouped_mm_16bit(sglk, dev, dt, rows, N, K, with_bias):
    g = torch.Generator().manual_seed(N + K + sum(rows))
    E, total = len(rows), sum(rows)
    act = (torch.randn(total, K, generator=g) * 0.1).to(dt)
    w = (torch.randn(E, N, K, generator=g) * 0.1).to(dt)
    bias = torch.randn(E, N, generator=g) * 0.005 if with_bias else None
    r = torch.tensor(rows, dtype=torch.int32)
    out = torch.empty(total, N, dtype=dt, device=dev)
    torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20(out, act.to(dev), w.to(dev), bias.to(dev) if with_bias else None, r.to(dev),
                                               E, 0, False, 1.702, 7.0)
    ref = omoe.moe_grouped_mm(act, w, bias, r)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=2e-2, atol=2e-3)


def test_grouped_mm_many_experts_tile_pipeline(sglk, dev):
    """moe_persist.hip beyond one wave of experts (72: two passes of the 64-lane prefix sums), with empty experts, one-row
    experts, a 3000-row expert, and the shortest K it takes (two 64-deep blocks); 16-bit and int4 weights."""
    g = torch.Generator().manual_seed(72)
    E, N, K, dt = 72, 256, 128, torch.bfloat16
    rows = [200 + 7 * (i % 9) for i in range(E)]
    rows[3], rows[40], rows[65], rows[70] = 0, 1, 3000, 0
    total = sum(rows)
    act = (torch.randn(total, K, generator=g) * 0.1).to(dt)
    r = torch.tensor(rows, dtype=torch.int32)
    w = (torch.randn(E, N, K, generator=g) * 0.1).to(dt)
    out = torch.full((total, N), float("nan"), dtype=dt, device=dev)
    torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20(out, act.to(dev), w.to(dev), None, r.to(dev), E, 0, False, 1.702, 7.0)
    torch.testing.assert_close(out.cpu().float(), omoe.moe_grouped_mm(act, w, None, r).float(), rtol=2e-2, atol=2e-3)
    packed, scales, _ = make_int4(E, N, K, 128, dt, False, g)
    out4 = torch.full((total, N), float("nan"), dtype=dt, device=dev)
    torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(out4, act.to(dev), packed.to(dev), scales.to(dev), None, None, r.to(dev), E,
                                                      True, 128)
    ref4 = omoe.moe_grouped_mm_w4a16(act, packed, scales, None, None, r, 128)
    torch.testing.assert_close(out4.cpu(), ref4, rtol=5e-2, atol=2e-2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_grouped_mm_bias_enters_exactly_on_the_tile_pipeline(sglk, dev, dt):
    """moe_persist.hip feeds the fp32 bias through the matrix pipe as three 16-bit pieces (hi + mid + lo): with zero
    activations the output must be the bias rounded ONCE to the output type, bit for bit - 256- and 128-row blocks, the
    streaming kernels' remainders, 16-bit and int4 weights, magnitudes over 30 binades."""
    g = torch.Generator().manual_seed(11)
    for rows in ([300, 200, 513, 256, 260, 250, 300, 310], [130, 100, 150, 120, 97, 140, 128, 111]):
        E, N, K = len(rows), 384, 256
        total = sum(rows)
        act = torch.zeros(total, K, dtype=dt)
        w = (torch.randn(E, N, K, generator=g) * 0.1).to(dt)
        mag = 2.0 ** torch.randint(-20, 10, (E, N), generator=g).float()
        bias = torch.randn(E, N, generator=g) * mag
        if dt == torch.float16:
            bias = bias.clamp(-6e4, 6e4)
        r = torch.tensor(rows, dtype=torch.int32)
        out = torch.full((total, N), float("nan"), dtype=dt, device=dev)
        torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20(out, act.to(dev), w.to(dev), bias.to(dev), r.to(dev), E, 0, False, 1.702, 7.0)
        want = torch.cat([bias[e].to(dt).expand(n, N) for e, n in enumerate(rows)])
        bad = (out.cpu() != want).nonzero()
        assert bad.numel() == 0, (rows[0], dt, bad.shape[0], [(int(i), int(j), float(out[i, j]), float(want[i, j]),
                                                                 float(bias[[sum(rows[:e + 1]) > int(i) for e in range(E)].index(True), j]))
                                                                for i, j in bad[:6]])
        packed, scales, _ = make_int4(E, N, K, 128, dt, False, g)
        out4 = torch.full((total, N), float("nan"), dtype=dt, device=dev)
        torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(out4, act.to(dev), packed.to(dev), scales.to(dev), None, bias.to(dev),
                                                          r.to(dev), E, True, 128)
        assert torch.equal(out4.cpu(), want), (rows[0], dt, "int4")


def test_grouped_mm_16bit_fused_act(sglk, dev):
    g = torch.Generator().manual_seed(5)
    _check_fused_act_16bit(sglk, dev, g, [5, 0, 9, 1, 40, 2, 2, 7], 256, 512)
    _check_fused_act_16bit(sglk, dev, g, [150, 99, 260, 128, 0, 131, 100, 177], 608, 384)  # eight-wave tiles, ragged N / 2
    _check_fused_act_16bit(sglk, dev, g, [300, 200, 513, 256, 260, 250, 300, 310], 640, 384)  # moe_persist.hip (no bias)


def _check_fused_act_16bit(sglk, dev, g, rows, N, K):
    E, dt = len(rows), torch.bfloat16
    act = (torch.randn(sum(rows), K, generator=g) * 0.1).to(dt)
    w = (torch.randn(E, N, K, generator=g) * 0.1).to(dt)
    r = torch.tensor(rows, dtype=torch.int32)
    bias = torch.randn(E, N, generator=g) * 0.01
    # (swiglu_gpt_oss - activation_type 2 with fuse_act, gate / up rows interleaved, reference moe_kernel.hpp:109-125 - with a limit
    #  small enough to bite on these inputs)
    for act_type, name in ((0, "silu"), (1, "gelu"), (3, "relu2"), (2, "swiglu_gpt_oss")):
        for b in (None, bias):
            out = torch.full((sum(rows), N if name == "relu2" else N // 2), float("nan"), dtype=dt, device=dev)
            torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20(out, act.to(dev), w.to(dev), b.to(dev) if b is not None else None,
                                                       r.to(dev), E, act_type, True, 1.702, 0.4)
            ref = omoe.moe_grouped_mm_fused(act, w, b, r, name, 1.702, 0.4)
            torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("T,topk,E,H,I", [(1, 2, 8, 1024, 512), (33, 6, 8, 1024, 1024), (222, 2, 64, 1024, 512),
                                          (64, 1, 8, 4096, 512), (64, 8, 128, 2816, 176)])
@pytest.mark.parametrize("activation,bias", [("silu", None), ("silu", "float32"), ("gelu", "bfloat16"), ("relu2", None)])
def test_fused_experts_16bit(sglk, dev, T, topk, E, H, I, activation, bias):
    """shapes / options of reference tests/test_moe_gemm.py:140-166 (bf16 weights, optional bf16 / fp32 bias,
    routed_scaling_factor 2.5)"""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(T + H + I)
    gate = 1 if activation == "relu2" else 2
    x = (torch.randn(T, H, generator=g) * 0.1).to(dt)
    w1 = (torch.randn(E, gate * I, H, generator=g) * 0.1).to(dt)
    w2 = (torch.randn(E, H, I, generator=g) * 0.1).to(dt)
    b1 = b2 = None
    if bias:
        bdt = torch.bfloat16 if bias == "bfloat16" else torch.float32
        b1 = (torch.randn(E, gate * I, generator=g) * 0.005).to(bdt)
        b2 = (torch.randn(E, H, generator=g) * 0.005).to(bdt)
    score = torch.softmax(torch.randn(T, E, generator=g).to(dt).float(), dim=-1)
    tw, ids = torch.topk(score, topk)
    d = lambda t: t.to(dev) if t is not None else None
    out = sglk.fused_experts(d(x), d(w1), d(w2), d(tw), d(ids), d(b1), d(b2), activation=activation, routed_scaling_factor=2.5)
    ref = omoe.fused_experts_16bit(x, w1, w2, tw, ids, b1, b2, activation, 2.5, fused_epilogue=True)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=3e-2, atol=1e-2)


@pytest.mark.parametrize("T,topk,E,H,I", [(1, 2, 8, 1024, 512), (4, 1, 8, 1024, 1024), (33, 6, 8, 1024, 1024),
                                          (222, 2, 64, 1024, 512), (64, 1, 8, 4096, 512), (222, 6, 8, 1024, 4096)])
@pytest.mark.parametrize("bias", [None, "bfloat16", "float32"])
def test_fused_experts_gpt_oss_swiglu(sglk, dev, T, topk, E, H, I, bias):
    """the ("silu", SWIGLU_ALPHA, SWIGLU_LIMIT) rows of reference tests/test_moe_gemm.py:141-160: gate / up interleaved in
    w1's rows, gate = min(gate, 7), up = clamp(up, -7, 7), gate * sigmoid(1.702 gate) * (up + 1)"""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(T + H + I + topk)
    x = (torch.randn(T, H, generator=g) * 0.5).to(dt)
    w1 = (torch.randn(E, 2 * I, H, generator=g) * 0.1).to(dt)  # (pre-activations of +-10: both clamps take part)
    w2 = (torch.randn(E, H, I, generator=g) * 0.03).to(dt)
    b1 = b2 = None
    if bias:
        bdt = torch.bfloat16 if bias == "bfloat16" else torch.float32
        b1 = (torch.randn(E, 2 * I, generator=g) * 0.5).to(bdt)
        b2 = (torch.randn(E, H, generator=g) * 0.005).to(bdt)
    score = torch.softmax(torch.randn(T, E, generator=g).to(dt).float(), dim=-1)
    tw, ids = torch.topk(score, topk)
    d = lambda t: t.to(dev) if t is not None else None
    out = sglk.fused_experts(d(x), d(w1), d(w2), d(tw), d(ids), d(b1), d(b2), activation="silu", routed_scaling_factor=2.5,
                             gemm1_alpha=1.702, gemm1_limit=7.0)
    ref = omoe.fused_experts_16bit(x, w1, w2, tw, ids, b1, b2, "silu", 2.5, gemm1_alpha=1.702, gemm1_limit=7.0)
    plain = omoe.fused_experts_16bit(x, w1, w2, tw, ids, b1, b2, "silu", 2.5)
    assert (ref.float() - plain.float()).abs().max() > 0.05  # (the case tells the gpt-oss form from the split-halves silu)
    # (a STRESS case next to test_fused_experts_gpt_oss_swiglu_reference_inputs, which runs the reference's inputs at the
    # reference's tolerance) rtol 3e-2, atol 1e-2 for outputs of magnitude <= 1; here they reach +-10 at I = 4096 and
    # an ulp of the bf16 intermediate [rows, 2I] (rounded once on both sides, summed in a different order) moves an output by
    # ~0.4 % of that range, so the absolute part scales with the range
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=3e-2, atol=1e-2 * max(1.0, ref.float().abs().max().item()))


@pytest.mark.parametrize("T,topk,E,H,I", [(1, 1, 8, 1024, 512), (33, 2, 8, 1024, 1024), (64, 6, 64, 1024, 512),
                                          (222, 2, 8, 4096, 512), (222, 6, 64, 1024, 4096)])
@pytest.mark.parametrize("bias", [None, "bfloat16", "float32"])
def test_fused_experts_gpt_oss_swiglu_reference_inputs(sglk, dev, T, topk, E, H, I, bias):
    """the same rows of reference tests/test_moe_gemm.py:141-160 with the reference's OWN input distribution (:23-24, :191-205:
    activations and weights N(0, 0.01), biases N(0, 0.005), scores softmax of N(0, 1)) and its OWN tolerance (:190, :237:
    rtol 1e-4, atol 1e-3) - the stress case above scales the inputs up until both clamps take part and needs a wider one."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(7 * T + H + I + topk + E)
    x = (torch.randn(T, H, generator=g) * 0.01).to(dt)
    w1 = (torch.randn(E, 2 * I, H, generator=g) * 0.01).to(dt)
    w2 = (torch.randn(E, H, I, generator=g) * 0.01).to(dt)
    b1 = b2 = None
    if bias:
        bdt = torch.bfloat16 if bias == "bfloat16" else torch.float32
        b1 = (torch.randn(E, 2 * I, generator=g) * 0.005).to(bdt)
        b2 = (torch.randn(E, H, generator=g) * 0.005).to(bdt)
    score = torch.softmax(torch.randn(T, E, generator=g).to(dt).float(), dim=-1)
    tw, ids = torch.topk(score, topk)
    d = lambda t: t.to(dev) if t is not None else None
    out = sglk.fused_experts(d(x), d(w1), d(w2), d(tw), d(ids), d(b1), d(b2), activation="silu", routed_scaling_factor=2.5,
                             gemm1_alpha=1.702, gemm1_limit=7.0)
    ref = omoe.fused_experts_16bit(x, w1, w2, tw, ids, b1, b2, "silu", 2.5, gemm1_alpha=1.702, gemm1_limit=7.0)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("explicit_zero", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [40, 260, 520])  # (decode: the reference's sequence; 65 rows per expert: the swiglu in GEMM 1's epilogue on
                                               #  the streaming kernels with the gather; 130: on the tile pipeline, K split in GEMM 2)
def test_fused_experts_int4_gpt_oss_swiglu(sglk, dev, explicit_zero, dtype, T):
    g = torch.Generator().manual_seed(5 + explicit_zero + T)
    E, topk, H, I, gs = 8, 2, 512, 256, 64
    x = torch.randn(T, H, generator=g).to(dtype)
    w1, s1, z1 = make_int4(E, 2 * I, H, gs, dtype, explicit_zero, g)
    w2, s2, z2 = make_int4(E, H, I, gs, dtype, explicit_zero, g)
    tw = torch.rand(T, topk, generator=g)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(T)]).to(torch.int32)
    d = lambda t: t.to(dev) if t is not None else None
    out = sglk.fused_experts(d(x), d(w1), d(w2), d(tw), d(ids), use_int4_w4a16=True, w1_scale=d(s1), w2_scale=d(s2),
                             w1_zp=d(z1), w2_zp=d(z2), gemm1_alpha=1.702, gemm1_limit=7.0)
    ref = omoe.fused_experts_int4(x, w1, w2, tw, ids, s1, s2, z1, z2, gemm1_alpha=1.702, gemm1_limit=7.0)
    # reference tolerance (:386: rtol 5e-2, atol 2e-2), its absolute part scaled to the output range (+-17 here) as above
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=5e-2, atol=1e-2 * max(2.0, ref.float().abs().max().item()))


def test_errors(sglk, dev):
    with pytest.raises(AssertionError):
        sglk.fused_experts(torch.zeros(1, 128, device=dev), torch.zeros(1, 2, 64, device=dev),
                           torch.zeros(1, 128, 1, device=dev), torch.ones(1, 1, device=dev),
                           torch.zeros(1, 1, dtype=torch.int32, device=dev))
    with pytest.raises(RuntimeError, match="up to 256"):
        sglk.topk_softmax(torch.empty(1, 1, device=dev), torch.empty(1, 1, dtype=torch.int32, device=dev),
                          torch.zeros(1, 300, dtype=torch.bfloat16, device=dev), False)
    with pytest.raises(RuntimeError, match="group_size"):
        torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(
            torch.empty(2, 8, dtype=torch.bfloat16, device=dev), torch.zeros(2, 96, dtype=torch.bfloat16, device=dev),
            torch.zeros(1, 8, 48, dtype=torch.uint8, device=dev), torch.ones(1, 8, 2, dtype=torch.bfloat16, device=dev),
            None, None, torch.tensor([2], dtype=torch.int32, device=dev), 1, True, 48)
