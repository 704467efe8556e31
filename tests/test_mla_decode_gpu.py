"""GPU parity: flash_mla_decode vs the CPU oracle. Input construction follows reference
tests/test_flash_mla_decode.py:62-146 (q x 100, random possibly repeated block table, q_nope made
from a transposed buffer, varlen sequences, page sizes 16..128, heads 16..128, splits -1 / 1)."""
import itertools

import pytest
import torch
from conftest import load_golden

from oracle import mla as omla

pytestmark = pytest.mark.gpu


def run_case(sglk, dev, dtype, seqs, page, H, splits, seed=42, q_scale=100.0, contiguous_q=False, table_tokens=0):
    g = torch.Generator().manual_seed(seed)
    bs = len(seqs)
    seq_lens = torch.tensor(seqs, dtype=torch.int32)
    block_num = (max(max(seqs), 1) + page - 1) // page
    pack = 128 // page
    block_num = (block_num + pack - 1) // pack * pack
    q = (torch.randn(bs, H, 576, generator=g) * q_scale).to(dtype)
    # table_tokens: a page table wider than any sequence (the launch is sized - splits, and for H > 64 the MFMA shape of
    # QK^T - by the table's width, the work by seq_lens)
    table = torch.randint(0, bs * block_num, (bs, max(block_num, (table_tokens + 127) // 128 * pack)), generator=g, dtype=torch.int32)
    cache = torch.randn(bs * block_num, page, 576, generator=g).to(dtype)
    scale = (128 + 64) ** -0.5
    ref = omla.mla_decode(q, cache, scale, table, seq_lens)

    qd = q.to(dev)
    if contiguous_q:
        q_nope = qd[:, :, :512].contiguous()
    else:  # as the reference test: a [H, bs, 512] buffer viewed as [bs, H, 512]
        q_nope = torch.empty((H, bs, 512), dtype=dtype, device=dev).transpose(0, 1)
        q_nope.copy_(qd[:, :, :512])
    q_pe = qd[:, :, 512:].clone()
    ws_size = sglk.flash_mla_get_workspace_size(table.shape[1] * page, bs, H, page, num_kv_splits=splits)
    ws = torch.empty(ws_size, device=dev, dtype=torch.uint8)
    out = sglk.flash_mla_decode(q_nope, q_pe, cache.to(dev), seq_lens.to(dev), table.to(dev), ws, scale, splits)
    assert out.shape == (bs, H, 512) and out.dtype == dtype
    atol, rtol = (1e-2, 1e-2) if dtype == torch.bfloat16 else (1e-3, 1e-3)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=atol, rtol=rtol)


def seqs_for(mean, bs, varlen, seed):
    if not varlen:
        return [mean] * bs
    g = torch.Generator().manual_seed(seed)
    s = torch.empty(bs).normal_(mean, mean / 2, generator=g).clip(2).to(torch.int32)
    return s.tolist()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mean", [128, 1024, 4096])
@pytest.mark.parametrize("bs", [1, 2, 4])
@pytest.mark.parametrize("varlen", [True, False])
@pytest.mark.parametrize("page", [16, 32, 64, 128])
@pytest.mark.parametrize("H", [16, 32, 64, 128])
def test_flash_mla_decode_grid(sglk, dev, dtype, mean, bs, varlen, page, H):
    """the reference's grid (tests/test_flash_mla_decode.py:62-71); its num_kv_splits axis (-1 / 1) alternates over it"""
    idx = (mean // 128) + bs + int(varlen) + (page // 16) + (H // 16) + (dtype == torch.float16)
    run_case(sglk, dev, dtype, seqs_for(mean, bs, varlen, idx), page, H, [-1, 1][idx % 2], seed=idx)


@pytest.mark.parametrize("seq", [1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 1000])
@pytest.mark.parametrize("page", [16, 64])
def test_tail_lengths(sglk, dev, seq, page):
    run_case(sglk, dev, torch.bfloat16, [seq, max(1, seq // 2), seq + 7], page, 32, 1, seed=seq)
    run_case(sglk, dev, torch.float16, [seq], page, 16, 4, seed=seq + 1)


@pytest.mark.parametrize("H", [1, 8, 17, 48, 100, 128])
def test_head_counts(sglk, dev, H):
    run_case(sglk, dev, torch.bfloat16, [300, 77], 32, H, -1, seed=H)


@pytest.mark.parametrize("splits", [2, 3, 8, 64, 128])
def test_explicit_splits(sglk, dev, splits):
    # more splits than tiles, empty splits, uneven sequences
    run_case(sglk, dev, torch.bfloat16, [700, 40, 1, 2048], 64, 64, splits, seed=splits)


def test_moderate_logits_exercise_softmax_mixing(sglk, dev):
    # q x 100 makes the softmax nearly one-hot; unit-scale q makes every token contribute
    run_case(sglk, dev, torch.bfloat16, [513, 2000], 64, 128, -1, q_scale=1.0)
    run_case(sglk, dev, torch.float16, [513, 2000], 128, 32, 1, q_scale=3.0, contiguous_q=True)


def test_running_max_rescale_is_forced(sglk, dev):
    """A key far along the sequence whose logit dwarfs everything before it forces the O-rescale branch late
    (guide rule: a rare data-dependent branch needs an input that takes it)."""
    dtype, H, page, n = torch.bfloat16, 32, 64, 1024
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, H, 576, generator=g).to(dtype)
    cache = torch.randn(n // page, page, 576, generator=g).to(dtype)
    table = torch.arange(n // page, dtype=torch.int32).view(1, -1)
    for pos in (40, 700, 1023):
        cache.view(-1, 576)[pos] = (q[0, 3].float() * 4).to(dtype)  # aligned with head 3 -> huge logit
    seq_lens = torch.tensor([n], dtype=torch.int32)
    scale = 576 ** -0.5
    ref = omla.mla_decode(q, cache, scale, table, seq_lens)
    for splits in (1, 4):
        ws = torch.empty(sglk.flash_mla_get_workspace_size(n, 1, H, page, splits), device=dev, dtype=torch.uint8)
        out = sglk.flash_mla_decode(q[..., :512].to(dev), q[..., 512:].to(dev).contiguous(), cache.to(dev),
                                    seq_lens.to(dev), table.to(dev), ws, scale, splits)
        torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mean", [128, 1024])
@pytest.mark.parametrize("bs", [1, 4])
@pytest.mark.parametrize("page", [16, 32, 64, 128])
@pytest.mark.parametrize("H", [96, 128])
def test_wide_table_16_wide_qk(sglk, dev, dtype, mean, bs, page, H):
    """H > 64 launches that keep every CU busy for long (>= 48 tiles per CU by the table's width) run QK^T on
    v_mfma_f32_16x16x32 (csrc/mla_decode.hip, S4): the reference grid's varlen cases behind a page table of 400k tokens -
    ragged tails, every page size, empty splits, both dtypes on that path."""
    idx = (mean // 128) + bs + (page // 16) + (H // 16) + (dtype == torch.float16)
    run_case(sglk, dev, dtype, seqs_for(mean, bs, True, idx), page, H, [-1, 1, 3][idx % 3], seed=idx, table_tokens=400_000 // bs * 1)


def test_full_size_config_sampled(sglk, dev):
    """BASELINE configs[3]: bs=128, seq=8192, H=128, page 64. The CPU oracle checks a sample of batch rows
    (each batch element is independent); every output must be finite."""
    dtype, bs, H, page, seq = torch.bfloat16, 128, 128, 64, 8192
    g = torch.Generator(device="cpu").manual_seed(1)
    n_pages = seq // page
    q = (torch.randn(bs, H, 576, generator=g) * 100).to(dtype)
    cache = torch.randn(bs * n_pages, page, 576, generator=g, dtype=torch.float32).to(dtype)
    table = torch.randint(0, bs * n_pages, (bs, n_pages), generator=g, dtype=torch.int32)
    seq_lens = torch.full((bs,), seq, dtype=torch.int32)
    scale = 576 ** -0.5
    ws = torch.empty(sglk.flash_mla_get_workspace_size(seq, bs, H, page, -1), device=dev, dtype=torch.uint8)
    qd = q.to(dev)
    out = sglk.flash_mla_decode(qd[..., :512], qd[..., 512:].contiguous(), cache.to(dev), seq_lens.to(dev),
                                table.to(dev), ws, scale, -1).cpu()
    assert torch.isfinite(out.float()).all()
    for i in (0, 57, 127):
        ref = omla.mla_decode(q[i:i + 1], cache, scale, table[i:i + 1], seq_lens[i:i + 1])
        torch.testing.assert_close(out[i:i + 1].float(), ref.float(), atol=1e-2, rtol=1e-2)


def test_golden_vectors(sglk, dev):
    for c in load_golden("mla_decode"):
        q, cache, table, seq_lens = c["q"], c["cache"], c["table"], c["seq_lens"]
        bs, H, _ = q.shape
        page = cache.shape[1]
        qd = q.to(dev)
        for splits in (1, -1):
            ws = torch.empty(sglk.flash_mla_get_workspace_size(table.shape[1] * page, bs, H, page, splits),
                             device=dev, dtype=torch.uint8)
            out = sglk.flash_mla_decode(qd[..., :512], qd[..., 512:].contiguous(), cache.to(dev), seq_lens.to(dev),
                                        table.to(dev), ws, c["scale"], splits)
            tol = 1e-2 if q.dtype == torch.bfloat16 else 1e-3
            torch.testing.assert_close(out.cpu().float(), c["out"].float(), atol=tol, rtol=tol)


def test_errors(sglk, dev):
    q_nope = torch.zeros(2, 16, 512, dtype=torch.bfloat16, device=dev)
    q_pe = torch.zeros(2, 16, 64, dtype=torch.bfloat16, device=dev)
    cache = torch.zeros(8, 64, 576, dtype=torch.bfloat16, device=dev)
    seq = torch.tensor([5, 6], dtype=torch.int32, device=dev)
    table = torch.zeros(2, 2, dtype=torch.int32, device=dev)
    ws = torch.empty(0, dtype=torch.uint8, device=dev)
    with pytest.raises(AssertionError):
        sglk.flash_mla_decode(q_nope, q_pe, cache, seq.long(), table, ws, 1.0, 1)
    with pytest.raises(AssertionError):
        sglk.flash_mla_decode(q_nope, q_pe, cache, seq, table[:, :1], ws, 1.0, 1)  # block_num % (128/page)
    with pytest.raises(RuntimeError, match="workspace too small"):
        sglk.flash_mla_decode(q_nope, q_pe, cache, seq, table, ws, 1.0, 2)
    with pytest.raises(RuntimeError, match="Unsupported page size"):
        torch.ops.sgl_kernel.flash_mla_decode(torch.empty_like(q_nope), q_nope, q_pe,
                                              torch.zeros(8, 48, 576, dtype=torch.bfloat16, device=dev), seq, table, ws,
                                              1.0, 1)
