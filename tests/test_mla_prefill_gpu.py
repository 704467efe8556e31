"""GPU parity: flash_mla_prefill vs the CPU oracle. Parameter matrix of reference
tests/test_flash_mla_prefill.py:96-160 (incremental and full prefill, ragged q, unaligned q / k, page sizes 16..128,
16 and 128 heads), plus non-causal and head counts the decode tests sweep."""
import pytest
import torch
from conftest import load_golden

from oracle import mla as omla

pytestmark = pytest.mark.gpu

CASES = [
    ([16], [128]), ([17], [128]), ([33], [256]), ([64], [256]), ([32, 32], [256, 256]),
    ([33, 33, 33, 33], [256, 256, 256, 256]), ([64, 16, 32], [256, 256, 256]), ([32, 16, 48], [128, 64, 256]),
    ([48, 16, 32], [256, 128, 512]), ([16, 16, 16, 16], [256, 64, 512, 128]), ([128, 16, 64], [256, 256, 256]),
    ([50, 17, 33], [512, 128, 256]), ([31, 1, 47, 15], [256, 64, 512, 128]), ([32, 16], [200, 100]),
    ([33, 17], [137, 71]), ([31, 1, 15], [250, 50, 150]), ([64], [64]), ([17, 17], [17, 17]),
    ([128, 16, 50], [128, 16, 50]), ([256, 256, 256, 256], [2048, 2048, 2048, 2048]), ([200, 200], [1536, 1536]),
    ([511], [2048]),
]


def run_case(sglk, dev, dtype, page, H, sqs, sks, causal=True, seed=42):
    g = torch.Generator().manual_seed(seed)
    bs = len(sqs)
    cu = torch.tensor([0] + torch.cumsum(torch.tensor(sqs), 0).tolist(), dtype=torch.int32)
    sk = torch.tensor(sks, dtype=torch.int32)
    block_num = (max(sks) + page - 1) // page
    pack = 128 // page
    block_num = (block_num + pack - 1) // pack * pack
    total_q = sum(sqs)
    qn = torch.randn(total_q, H, 512, generator=g).to(dtype)
    qp = torch.randn(total_q, H, 64, generator=g).to(dtype)
    table = torch.randint(0, bs * block_num, (bs, block_num), generator=g, dtype=torch.int32)
    cache = torch.randn(int(table.max()) + 1, page, 576, generator=g).to(dtype)
    scale = (128 + 64) ** -0.5
    ref = omla.mla_prefill(qn, qp, cache, scale, table, cu, sk, causal=causal)
    ws = torch.empty(sglk.flash_mla_prefill_get_workspace_size(block_num * page, bs), device=dev, dtype=torch.uint8)
    out = sglk.flash_mla_prefill(qn.to(dev), qp.to(dev), cache.to(dev), cu.to(dev), sk.to(dev), max(sqs), table.to(dev),
                                 ws, scale, causal=causal, num_kv_splits=1)
    assert out.shape == (total_q, H, 512) and out.dtype == dtype
    atol, rtol = (1e-2, 1e-2) if dtype == torch.bfloat16 else (1e-3, 1e-3)  # reference tolerance (:235-236)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("idx", range(len(CASES)))
@pytest.mark.parametrize("H", [16, 128])
@pytest.mark.parametrize("page", [16, 32, 64, 128])
def test_mla_prefill_reference_matrix(sglk, dev, idx, H, page):
    """the reference's matrix (tests/test_flash_mla_prefill.py:98-140: block sizes x head counts x length cases); its dtype
    axis alternates over it"""
    sqs, sks = CASES[idx]
    dtype = [torch.bfloat16, torch.float16][(idx + page // 16 + H // 16) % 2]
    run_case(sglk, dev, dtype, page, H, sqs, sks, seed=idx)


@pytest.mark.parametrize("H", [1, 8, 17, 32, 48, 64, 100])
def test_mla_prefill_head_counts(sglk, dev, H):
    run_case(sglk, dev, torch.bfloat16, 64, H, [37, 5, 64], [300, 5, 64], seed=H)


@pytest.mark.parametrize("H", [16, 32, 64, 100, 128])
def test_mla_prefill_16_wide_qk_head_slots(sglk, dev, H):
    """grids of at least one workgroup per CU run QK^T on v_mfma_f32_16x16x32 (csrc/mla_decode.hip, S4): every (token slot,
    head) row layout of a 128-row workgroup on that path, ragged q, unaligned k, causal and not"""
    tpw = 128 // (16 if H <= 16 else 32 if H <= 32 else 64 if H <= 64 else 128)
    sq = 64 * tpw + 3
    run_case(sglk, dev, [torch.bfloat16, torch.float16][H // 16 % 2], 64, H, [sq, sq - 5, sq, 7], [sq + 190, sq + 64, sq, 333], seed=H)
    run_case(sglk, dev, torch.bfloat16, 16, H, [sq, sq, 9, sq], [sq + 50, sq + 17, 100, sq + 1], causal=False, seed=H + 1)


def test_mla_prefill_non_causal(sglk, dev):
    run_case(sglk, dev, torch.float16, 32, 16, [33, 7], [137, 71], causal=False)
    run_case(sglk, dev, torch.bfloat16, 128, 64, [5], [1000], causal=False)


def test_mla_prefill_medium_32_heads(sglk, dev):
    run_case(sglk, dev, torch.bfloat16, 64, 32, [200, 130], [1536, 700])


def test_mla_prefill_golden_vectors(sglk, dev):
    for c in load_golden("mla_prefill"):
        qn = c["q_nope"]
        sqs = (c["cu_seqlens_q"][1:] - c["cu_seqlens_q"][:-1]).tolist()
        ws = torch.empty(0, device=dev, dtype=torch.uint8)
        out = sglk.flash_mla_prefill(qn.to(dev), c["q_pe"].to(dev), c["cache"].to(dev), c["cu_seqlens_q"].to(dev),
                                     c["seq_lens_k"].to(dev), max(sqs), c["table"].to(dev), ws, c["scale"], True, 1)
        tol = 1e-2 if qn.dtype == torch.bfloat16 else 1e-3
        torch.testing.assert_close(out.cpu().float(), c["out"].float(), atol=tol, rtol=tol)


def test_mla_prefill_equals_decode_for_single_token(sglk, dev):
    """One new token per sequence is a decode step: both entry points must agree bit for bit at splits = 1."""
    g = torch.Generator().manual_seed(3)
    bs, H, page, n = 3, 16, 64, 256
    q = torch.randn(bs, H, 576, generator=g).to(torch.bfloat16).to(dev)
    cache = torch.randn(bs * 4, page, 576, generator=g).to(torch.bfloat16).to(dev)
    table = torch.randint(0, bs * 4, (bs, 4), generator=g, dtype=torch.int32).to(dev)
    sk = torch.tensor([n, 100, 1], dtype=torch.int32, device=dev)
    cu = torch.arange(bs + 1, dtype=torch.int32, device=dev)
    ws = torch.empty(0, device=dev, dtype=torch.uint8)
    scale = 576 ** -0.5
    o_pre = sglk.flash_mla_prefill(q[..., :512].contiguous(), q[..., 512:].contiguous(), cache, cu, sk, 1, table, ws, scale)
    o_dec = sglk.flash_mla_decode(q[..., :512].contiguous(), q[..., 512:].contiguous(), cache, sk, table, ws, scale, 1)
    torch.testing.assert_close(o_pre.float(), o_dec.float(), atol=1e-2, rtol=1e-2)
