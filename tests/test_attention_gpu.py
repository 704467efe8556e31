"""GPU parity: `fwd` (flash_attn_with_kvcache / flash_attn_varlen_func) vs the CPU oracle.

Grids follow reference tests/test_flash_attention.py: paged kv-cache :626-658, decode :1173-1224, varlen
:1908-1948, out= :2249-2300. Acceptance is the reference's rule (:1112-1121): the error against the fp32 oracle
must not exceed twice the error of a plain torch implementation in the input dtype (+1e-5)."""
import itertools

import pytest
import torch
from conftest import load_golden

from oracle import attention as oa

pytestmark = pytest.mark.gpu


def pt_seq(q, k, v, scale, causal, window, softcap, sinks):
    """The same math in the input dtype (no upcast) — the reference's yardstick `out_pt`: scores, probabilities and the
    output rounded to the input dtype. (The products run as fp32 matmuls of the rounded operands and their results are
    rounded: what a 16-bit matmul with fp32 accumulation returns, without the CPU's slow half-precision GEMM.)"""
    dt = q.dtype
    sq, Hq, D = q.shape
    g = Hq // k.shape[1]
    kf = (k * scale).repeat_interleave(g, dim=1)
    scores = torch.einsum("thd,shd->hts", q.float(), kf.to(dt).float()).to(dt).float()
    if softcap > 0:
        scores = torch.tanh(scores / softcap) * softcap
    sk = k.shape[0]
    left, right = window
    if causal:
        right = 0
    row = torch.arange(sq).view(-1, 1)
    col = torch.arange(sk).view(1, -1)
    mask = torch.zeros(sq, sk, dtype=torch.bool)
    if right >= 0:
        mask |= col > row + sk - sq + right
    if left >= 0:
        mask |= col < row + sk - sq - left
    scores = scores.masked_fill(mask.unsqueeze(0), float("-inf"))
    if sinks is not None:
        scores = torch.cat([scores, sinks.float().view(Hq, 1, 1).expand(Hq, sq, 1)], dim=-1)
    attn = torch.nan_to_num(torch.softmax(scores, dim=-1), nan=0.0).to(dt)
    if sinks is not None:
        attn = attn[..., :-1]
    return torch.einsum("hts,shd->thd", attn.float(), v.repeat_interleave(g, dim=1).float()).to(dt).float()


def check(out, ref, pt, what=""):
    err = (out.float() - ref).abs().max().item()
    err_pt = (pt - ref).abs().max().item()
    assert err <= 2 * err_pt + 1e-5, f"{what}: max err {err:.3e} > 2 x {err_pt:.3e}"
    m = (out.float() - ref).abs().mean().item()
    m_pt = (pt - ref).abs().mean().item()
    assert m <= 1.5 * m_pt + 1e-6, f"{what}: mean err {m:.3e} > 1.5 x {m_pt:.3e}"


_CASE_CACHE = {}


def paged_case(dtype, seqs_q, seqs_k, Hq, Hk, D, causal, window, softcap, use_sink, seed):
    """Logical inputs + the CPU oracle and its low-precision yardstick for one ragged batch. Nothing here depends on the
    page size or the split count, so the (slow) CPU side is computed once per logical case and reused by every
    (page, num_splits) launch of it (one entry kept: the callers iterate page / splits innermost)."""
    key = (dtype, tuple(seqs_q), tuple(seqs_k), Hq, Hk, D, causal, tuple(window), softcap, use_sink, seed)
    hit = _CASE_CACHE.get(key)
    if hit is not None:
        return hit
    _CASE_CACHE.clear()
    g = torch.Generator().manual_seed(seed)
    b = len(seqs_q)
    q = torch.randn(sum(seqs_q), Hq, D, generator=g).to(dtype)
    cu_q = torch.tensor([0] + list(itertools.accumulate(seqs_q)), dtype=torch.int32)
    s_pad = (max(seqs_k) + 255) // 256 * 256  # whole pages for every page size used (32..256); the tail is random too
    k_log = torch.randn(b, s_pad, Hk, D, generator=g).to(dtype)
    v_log = torch.randn(b, s_pad, Hk, D, generator=g).to(dtype)
    sinks = torch.randn(Hq, generator=g) if use_sink else None
    scale = D ** -0.5
    ks = [k_log[i, :seqs_k[i]] for i in range(b)]
    vs = [v_log[i, :seqs_k[i]] for i in range(b)]
    ref, ref_lse = oa.attention_ragged(q, ks, vs, cu_q, scale, causal=causal, window=window, softcap=softcap, sinks=sinks)
    pt = torch.zeros_like(ref)
    for i in range(b):
        s, e = cu_q[i], cu_q[i + 1]
        if e > s:
            pt[s:e] = pt_seq(q[s:e], ks[i], vs[i], scale, causal, window, softcap, sinks)
    case = dict(q=q, cu_q=cu_q, k_log=k_log, v_log=v_log, sinks=sinks, scale=scale, ref=ref, ref_lse=ref_lse, pt=pt,
                dev={})
    _CASE_CACHE[key] = case
    return case


def run_paged(sglk, dev, dtype, seqs_q, seqs_k, Hq, Hk, D, page, causal=False, window=(-1, -1), softcap=0.0,
              use_sink=False, num_splits=0, seed=0, use_out=False):
    c = paged_case(dtype, seqs_q, seqs_k, Hq, Hk, D, causal, window, softcap, use_sink, seed)
    b = len(seqs_q)
    if page not in c["dev"]:  # cut the logical rows into shuffled pages (3 spare pages of noise), upload once per page size
        c["dev"].clear()
        g = torch.Generator().manual_seed(seed * 131 + page)
        pages_per_seq = (max(seqs_k) + page - 1) // page
        n_pages = b * pages_per_seq + 3
        table = torch.randperm(n_pages, generator=g)[: b * pages_per_seq].view(b, pages_per_seq)
        kc = torch.randn(n_pages, page, Hk, D, generator=g).to(dtype) if n_pages * page * Hk * D < 2**22 else \
            torch.zeros(n_pages, page, Hk, D, dtype=dtype)
        vc = kc.clone()
        kc[table.view(-1)] = c["k_log"][:, :pages_per_seq * page].reshape(b * pages_per_seq, page, Hk, D)
        vc[table.view(-1)] = c["v_log"][:, :pages_per_seq * page].reshape(b * pages_per_seq, page, Hk, D)
        assert torch.equal(oa.gather_paged(kc, table[b - 1], seqs_k[b - 1]), c["k_log"][b - 1, :seqs_k[b - 1]])
        c["dev"][page] = (c["q"].to(dev), kc.to(dev), vc.to(dev), torch.tensor(seqs_k, dtype=torch.int32, device=dev),
                          table.to(torch.int32).to(dev), c["cu_q"].to(dev), c["sinks"].to(dev) if use_sink else None)
    q_d, kc_d, vc_d, lens_d, table_d, cu_q_d, sinks_d = c["dev"][page]
    out_buf = torch.empty(q_d.shape, dtype=dtype, device=dev) if use_out else None
    res = sglk.flash_attn_with_kvcache(
        q_d, kc_d, vc_d, cache_seqlens=lens_d, page_table=table_d, cu_seqlens_q=cu_q_d, max_seqlen_q=max(seqs_q),
        softmax_scale=c["scale"], sinks=sinks_d, causal=causal, window_size=window, softcap=softcap,
        num_splits=num_splits, return_softmax_lse=True, out=out_buf)
    out, lse = res[0], res[1]
    if use_out:
        assert out.data_ptr() == out_buf.data_ptr()
    check(out.cpu(), c["ref"], c["pt"], f"paged D={D} page={page} splits={num_splits}")
    fin = torch.isfinite(c["ref_lse"])
    torch.testing.assert_close(lse.cpu()[fin], c["ref_lse"][fin], rtol=1e-3, atol=1e-3)


# ---------------------------------------------------------------------- paged kv-cache, mixed prefill/decode batches
@pytest.mark.parametrize("page", [64, 128])  # outermost decorator = fastest-varying: both page sizes reuse one oracle run
@pytest.mark.parametrize("heads", [(16, 16), (16, 4), (8, 1)])
@pytest.mark.parametrize("causal,local", [(False, True), (False, False), (True, False)])
@pytest.mark.parametrize("D", [64, 128, 256, 512])
@pytest.mark.parametrize("sq,sk", [(3, 1024), (64, 800), (64, 256), (3, 799), (64, 2048), (128, 128), (256, 512), (512, 512)])
def test_kvcache_paged(sglk, dev, heads, causal, local, page, D, sq, sk):
    """the reference's paged kv-cache grid (tests/test_flash_attention.py:70-82, 630-658: 9 (seqlen_q, seqlen_k) pairs x d
    64..512 x page 64 / 128 x masks x head layouts; the 2048 x 3577 pair is test_kvcache_paged_many_tiles below). Its dtype
    and sink axes alternate over the grid instead of doubling it."""
    Hq, Hk = heads
    par = sq + sk // 3 + Hk + 3 * causal + 2 * local + D // 64  # same for both page sizes
    dtype = torch.float16 if par % 2 else torch.bfloat16
    g = torch.Generator().manual_seed(sq + sk)
    seqs_k = torch.randint(max(1, sk - 20), sk + 1, (3,), generator=g).tolist()
    seqs_q = [min(sq, s) for s in seqs_k]
    window = (torch.randint(0, sk, (2,), generator=g).tolist()) if local else (-1, -1)
    run_paged(sglk, dev, dtype, seqs_q, seqs_k, Hq, Hk, D, page, causal=causal, window=tuple(window),
              use_sink=((par // 2) % 2 == 0), seed=sq + sk)


@pytest.mark.parametrize("page", [64, 128])
@pytest.mark.parametrize("causal,local", [(False, True), (False, False), (True, False)])
@pytest.mark.parametrize("D", [128, 512])
def test_kvcache_paged_many_tiles(sglk, dev, causal, local, page, D):
    """(2048, 3577): "enough tiles to test the persistent scheduler" (reference :81), two sequences"""
    window = (700, 300) if local else (-1, -1)
    run_paged(sglk, dev, torch.bfloat16, [2048, 1999], [3577, 3001], 16, 4, D, page, causal=causal, window=window,
              use_sink=causal, seed=D)


@pytest.mark.parametrize("page", [8, 16, 32, 256])
@pytest.mark.parametrize("D", [64, 96, 128, 192])  # (96 / 192: inside the 128 / 256 LDS images, round 5)
@pytest.mark.parametrize("causal,local", [(True, False), (False, True)])
def test_prefill_kernel_page_sizes(sglk, dev, causal, local, D, page):
    """prefill-sized row counts (the LDS-DMA kernel, round 4) over pages smaller than, equal to and larger than what one wave
    stages per tile (16 rows): pages of 8 tokens take the per-lane address path (a wave's rows straddle pages), 16 and up the
    scalar one; the ragged tails and the second sequence's 129 queries end inside tiles and pages."""
    window = (150, 40) if local else (-1, -1)
    run_paged(sglk, dev, torch.bfloat16 if D == 128 else torch.float16, [300, 129], [517, 640], 8, 2, D, page, causal=causal,
              window=window, seed=D + page)


@pytest.mark.parametrize("D,sq,sk", [(512, 3, 300), (512, 40, 130), (96, 17, 200), (192, 5, 77)])
def test_kvcache_other_head_dims(sglk, dev, D, sq, sk):
    run_paged(sglk, dev, torch.bfloat16, [sq, 1], [sk, sk - 9], 8, 2, D, 64, causal=True, seed=D)


# --------------------------------------------------------------------------------------------------- decode
@pytest.mark.parametrize("page", [64, 128])  # fastest-varying: the CPU oracle of a logical case runs once for both
@pytest.mark.parametrize("heads", [(16, 16), (16, 4), (16, 1), (8, 1), (32, 8)])
@pytest.mark.parametrize("local", [False, True])
@pytest.mark.parametrize("D", [64, 128, 256, 512])
@pytest.mark.parametrize("batch,seqlen_k", [(1, 1), (4, 63), (1, 64), (4, 65), (1, 129), (4, 512), (4, 1024), (1, 4033),
                                            (1, 4096), (2, 4097), (4, 4608), (1, 5120), (2, 8192)])
def test_decode(sglk, dev, heads, local, page, D, batch, seqlen_k):
    """the reference's decode grid (tests/test_flash_attention.py:1177-1222: 13 cache lengths x d 64..512 x local x head
    layouts x page 64 / 128, batch 1 and 4); its sink axis alternates over the grid, every case at three split counts"""
    Hq, Hk = heads
    g = torch.Generator().manual_seed(seqlen_k + D)
    seqs_k = torch.randint(max(1, seqlen_k - 30), seqlen_k + 1, (batch,), generator=g).tolist()
    seqs_k[0] = seqlen_k
    window = (seqlen_k // 3, 0) if local else (-1, -1)
    for splits in (0, 1, 5):
        run_paged(sglk, dev, torch.bfloat16, [1] * batch, seqs_k, Hq, Hk, D, page, window=window,
                  use_sink=((seqlen_k + (D >> 6) + Hk + int(local)) % 2 == 0), num_splits=splits, seed=seqlen_k)


@pytest.mark.parametrize("page", [64, 128])
@pytest.mark.parametrize("heads", [(16, 4), (8, 1), (32, 8)])
@pytest.mark.parametrize("local", [False, True])
@pytest.mark.parametrize("D", [96, 192])
@pytest.mark.parametrize("batch,seqlen_k", [(1, 1), (4, 65), (1, 129), (4, 1024), (1, 4033), (2, 4097)])
def test_decode_d96_d192(sglk, dev, heads, local, page, D, batch, seqlen_k):
    """head dims 96 / 192 of the reference's paged decode (FMHADecodeXe20.cmake:13-16) on the independent-wave decode kernel inside
    its d = 128 / 256 forms (round 5): a slice of the decode grid above, every case at three split counts"""
    Hq, Hk = heads
    g = torch.Generator().manual_seed(seqlen_k + D)
    seqs_k = torch.randint(max(1, seqlen_k - 30), seqlen_k + 1, (batch,), generator=g).tolist()
    seqs_k[0] = seqlen_k
    window = (seqlen_k // 3, 0) if local else (-1, -1)
    for splits in (0, 1, 5):
        run_paged(sglk, dev, torch.bfloat16 if D == 96 else torch.float16, [1] * batch, seqs_k, Hq, Hk, D, page, window=window,
                  use_sink=((seqlen_k + Hk + int(local)) % 2 == 0), num_splits=splits, seed=seqlen_k)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("heads,sq", [((16, 16), 1), ((16, 16), 5), ((16, 4), 4), ((16, 2), 2), ((32, 2), 1), ((8, 1), 2)])
@pytest.mark.parametrize("feature", ["plain", "causal", "local", "softcap", "sinks", "causal+sinks"])
def test_decode_kernel_features(sglk, dev, dtype, heads, sq, feature):
    """the independent-wave decode kernel (d = 128, at most 16 packed rows per kv head, pages >= 32 tokens): every mask /
    softcap / sink combination of the contract, 1..16 rows, ragged lengths around tile and page edges, all split counts"""
    Hq, Hk = heads
    seqs_k = [1, 31, 32, 33, 640, 1000 + sq]
    seqs_q = [min(sq, k) for k in seqs_k]
    kw = dict(causal="causal" in feature, window=(100, 0) if feature == "local" else (-1, -1),
              softcap=20.0 if feature == "softcap" else 0.0, use_sink="sinks" in feature)
    for page, splits in ((16, 0), (16, 5), (32, 0), (64, 1), (64, 16), (256, 3)):  # (16: a tile's two halves in two pages, round 5 late)
        run_paged(sglk, dev, dtype, seqs_q, seqs_k, Hq, Hk, 128, page, num_splits=splits, seed=sq + Hq, **kw)


@pytest.mark.parametrize("D", [64, 96, 128, 192, 256])
def test_decode_kernel_16_token_pages(sglk, dev, D):
    """pages of 16 tokens on the decode kernel at every head dim it has (a 32-token tile = two pages: two ids per tile), lengths that
    end in the first / second half of a tile, 1 .. 40 packed rows (fp8 caches over 16-token pages: test_fp8_kvcache)"""
    for sq, heads in ((1, (16, 4)), (3, (8, 8)), (10, (16, 4))):
        seqs_k = [1, 15, 16, 17, 47, 48, 49, 1000 + sq]
        run_paged(sglk, dev, torch.bfloat16, [min(sq, k) for k in seqs_k], seqs_k, heads[0], heads[1], D, 16, causal=True,
                  seed=D + sq)


def test_decode_full_size_config_sampled(sglk, dev):
    """BASELINE configs[2] decode: b=16, Hq=32, Hk=8, d=128, seq=4096, paged (64), bf16; oracle on 3 sequences."""
    b, Hq, Hk, D, S, page = 16, 32, 8, 128, 4096, 64
    g = torch.Generator().manual_seed(3)
    q = torch.randn(b, 1, Hq, D, generator=g).to(torch.bfloat16)
    n_pages = b * S // page
    kc = torch.randn(n_pages, page, Hk, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(n_pages, page, Hk, D, generator=g).to(torch.bfloat16)
    table = torch.randperm(n_pages, generator=g).view(b, -1).to(torch.int32)
    lens = torch.full((b,), S, dtype=torch.int32, device=dev)
    out = sglk.flash_attn_with_kvcache(q.to(dev), kc.to(dev), vc.to(dev), cache_seqlens=lens, page_table=table.to(dev)).cpu()
    assert out.shape == (b, Hq, D) and torch.isfinite(out.float()).all()
    for i in (0, 7, 15):
        k_i, v_i = oa.gather_paged(kc, table[i], S), oa.gather_paged(vc, table[i], S)
        ref, _ = oa.attention_seq(q[i], k_i, v_i, D ** -0.5)
        check(out[i:i + 1], ref, pt_seq(q[i], k_i, v_i, D ** -0.5, False, (-1, -1), 0.0, None), f"seq {i}")


# -------------------------------------------------------------------------------------------- ragged (non-paged)
@pytest.mark.parametrize("heads", [(16, 16), (16, 4), (16, 1)])
@pytest.mark.parametrize("causal,local", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("D", [72, 80, 128, 192, 256, 512])
@pytest.mark.parametrize("sq,sk", [(1, 1), (1, 3), (2, 1), (511, 1), (3, 513), (1, 239), (3, 799), (64, 128), (128, 128),
                                   (256, 256), (113, 203), (128, 217), (113, 211), (108, 256), (256, 512), (307, 256),
                                   (640, 128), (512, 256), (1024, 1024), (1023, 1024), (1024, 1023)])
def test_varlen(sglk, dev, heads, causal, local, D, sq, sk):
    """the reference's ragged grid (tests/test_flash_attention.py:1912-1947: head layouts x masks x d x 20 length pairs; the
    2048 x 2048 pair is test_varlen_2048 below); its softcap axis (0 / 15) alternates over the grid instead of doubling it"""
    run_varlen(sglk, dev, heads, causal, local, D, sq, sk)


@pytest.mark.parametrize("heads", [(16, 16), (16, 4), (16, 1)])
@pytest.mark.parametrize("causal,local", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("sq,sk", [(128, 128), (113, 203), (256, 512), (307, 256), (640, 128), (1023, 1024)])
def test_varlen_d96(sglk, dev, heads, causal, local, sq, sk):
    """head dim 96 (reference FMHAPrefillXe20.cmake:30-54) at prefill row counts: the 128-row-block kernel with 192-byte rows inside
    the d = 128 LDS image (d = 192 inside the d = 256 one is part of test_varlen's grid)"""
    run_varlen(sglk, dev, heads, causal, local, 96, sq, sk)


@pytest.mark.parametrize("causal,local", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("D", [80, 128, 256])
def test_varlen_2048(sglk, dev, causal, local, D):
    run_varlen(sglk, dev, (16, 4), causal, local, D, 2048, 2048)


def run_varlen(sglk, dev, heads, causal, local, D, sq, sk):
    Hq, Hk = heads
    softcap = 15.0 if (sq + sk + D // 8 + Hk) % 2 else 0.0
    dtype = torch.bfloat16 if (sq + D) % 2 else torch.float16
    g = torch.Generator().manual_seed(sq * 7 + sk + D)
    b = 3 if sq < 2048 else 2
    lens_q = torch.randint(max(1, sq - 20), sq + 1, (b,), generator=g).tolist()
    lens_k = torch.randint(max(1, sk - 20), sk + 1, (b,), generator=g).tolist()
    cu_q = torch.tensor([0] + list(itertools.accumulate(lens_q)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(itertools.accumulate(lens_k)), dtype=torch.int32)
    q = torch.randn(sum(lens_q), Hq, D, generator=g).to(dtype)
    k = torch.randn(sum(lens_k), Hk, D, generator=g).to(dtype)
    v = torch.randn(sum(lens_k), Hk, D, generator=g).to(dtype)
    window = tuple(torch.randint(0, sk, (2,), generator=g).tolist()) if local else (-1, -1)
    scale = D ** -0.5
    ks = [k[cu_k[i]:cu_k[i + 1]] for i in range(b)]
    vs = [v[cu_k[i]:cu_k[i + 1]] for i in range(b)]
    ref, _ = oa.attention_ragged(q, ks, vs, cu_q, scale, causal=causal, window=window, softcap=softcap)
    pt = torch.cat([pt_seq(q[cu_q[i]:cu_q[i + 1]], ks[i], vs[i], scale, causal, window, softcap, None) for i in range(b)])
    out = sglk.flash_attn_varlen_func(q.to(dev), k.to(dev), v.to(dev), cu_q.to(dev), cu_k.to(dev), max(lens_q),
                                      max(lens_k), causal=causal, window_size=window, softcap=softcap)
    check(out.cpu(), ref, pt, f"varlen D={D} softcap={softcap}")


def test_softcap_and_out_buffer(sglk, dev):
    run_paged(sglk, dev, torch.float16, [20, 1, 7], [40, 300, 64], 6, 2, 80, 64, causal=True, softcap=30.0, seed=1)
    run_paged(sglk, dev, torch.bfloat16, [1, 1], [500, 77], 16, 4, 128, 128, num_splits=4, use_out=True, seed=2)


def test_fully_masked_rows_give_zero(sglk, dev):
    # seqlen_q > seqlen_k with a causal mask: the first rows see no key (reference docstring example)
    q = torch.randn(5, 2, 64).to(torch.float16)
    k = torch.randn(2, 2, 64).to(torch.float16)
    v = torch.randn(2, 2, 64).to(torch.float16)
    cu_q = torch.tensor([0, 5], dtype=torch.int32)
    cu_k = torch.tensor([0, 2], dtype=torch.int32)
    out = sglk.flash_attn_varlen_func(q.to(dev), k.to(dev), v.to(dev), cu_q.to(dev), cu_k.to(dev), 5, 2, causal=True).cpu()
    assert torch.equal(out[:3], torch.zeros(3, 2, 64, dtype=torch.float16))
    ref, _ = oa.attention_seq(q, k, v, 64 ** -0.5, causal=True)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


def test_golden_vectors(sglk, dev):
    for c in load_golden("attention"):
        q, k, v = c["q"], c["k"], c["v"]
        b, sq, Hq, D = q.shape
        sk = k.shape[1]
        cu_q = torch.arange(0, b + 1, dtype=torch.int32) * sq
        cu_k = torch.arange(0, b + 1, dtype=torch.int32) * sk
        out = sglk.flash_attn_varlen_func(
            q.reshape(-1, Hq, D).to(dev), k.reshape(-1, k.shape[2], D).to(dev), v.reshape(-1, k.shape[2], D).to(dev),
            cu_q.to(dev), cu_k.to(dev), sq, sk, softmax_scale=c["scale"],
            sinks=c["sink"].to(dev) if c["sink"] is not None else None, causal=c["causal"], window_size=c["window"],
            softcap=c["softcap"]).cpu().view(b, sq, Hq, D)
        # the reference's own acceptance rule against its own fp32 and low-precision outputs
        err = (out.float() - c["out"].float()).abs().max().item()
        err_pt = (c["out_pt"].float() - c["out"].float()).abs().max().item()
        assert err <= 2 * err_pt + 1e-5, (err, err_pt)


def test_errors(sglk, dev):
    q = torch.zeros(4, 2, 64, dtype=torch.float16, device=dev)
    k = torch.zeros(8, 1, 64, dtype=torch.float16, device=dev)
    cu = torch.tensor([0, 4], dtype=torch.int32, device=dev)
    cuk = torch.tensor([0, 8], dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="q_v is not supported"):
        sglk.flash_attn_varlen_func(q, k, k, cu, cuk, 4, 8, qv=q)
    with pytest.raises(RuntimeError, match="same dtype"):
        sglk.flash_attn_varlen_func(q, k.to(torch.bfloat16), k, cu, cuk, 4, 8)
    with pytest.raises(RuntimeError, match="must divide"):
        sglk.flash_attn_varlen_func(q[:, :1].repeat(1, 3, 1), k.repeat(1, 2, 1), k.repeat(1, 2, 1), cu, cuk, 4, 8)


def test_missing_max_seqlen_q_is_safe(sglk, dev):
    """ragged q with max_seqlen_q left at 0 (the reference wrapper's default for cu_seqlens_q calls): d = 128 and 40 packed
    rows must not be handed to the 16-row decode kernel; the wrapper substitutes the row count as the bound."""
    g = torch.Generator().manual_seed(11)
    q = torch.randn(40, 4, 128, generator=g).to(torch.bfloat16)
    k = torch.randn(300, 2, 128, generator=g).to(torch.bfloat16)
    v = torch.randn(300, 2, 128, generator=g).to(torch.bfloat16)
    cu_q = torch.tensor([0, 20, 40], dtype=torch.int32)
    cu_k = torch.tensor([0, 100, 300], dtype=torch.int32)
    args = (q.to(dev), k.to(dev), v.to(dev), cu_q.to(dev), cu_k.to(dev))
    good = sglk.flash_attn_varlen_func(*args, 20, 200, causal=True)
    lazy = sglk.flash_attn_varlen_func(*args, 0, 200, causal=True)
    # (since round 5 the two bounds pick different kernels - 40 packed rows: the decode kernel's 16-row groups; 80: the 128-row-block
    #  kernel - so the two results agree to rounding, not bit for bit)
    torch.testing.assert_close(good.float(), lazy.float(), rtol=2e-2, atol=2e-2)
    ref, _ = oa.attention_ragged(q, [k[:100], k[100:]], [v[:100], v[100:]], cu_q, 128 ** -0.5, causal=True)
    torch.testing.assert_close(lazy.float().cpu(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(good.float().cpu(), ref, rtol=2e-2, atol=2e-2)


# ---------------------------------------------------------------------- fp8 KV cache (reference :1697-1830)
@pytest.mark.parametrize("fp8_dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("heads", [(8, 8), (8, 2)])
@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("page", [16, 64, 128])
@pytest.mark.parametrize("sq", [1, 32, 64, 200])
@pytest.mark.parametrize("sk", [256, 512])
@pytest.mark.parametrize("causal", [False, True])
def test_fp8_kvcache(sglk, dev, fp8_dtype, heads, D, page, sq, sk, causal):
    """bf16 q against an fp8 paged KV cache with per-tensor descales (scalar and expanded-scalar layouts). 64 queries x 4 q
    heads per kv head and 200 queries reach the 128-row-block prefill kernel at d = 64 / 128 (fp8 staging through registers,
    round 5; reference tests/test_flash_attention.py:1691-1704), the rest the decode and the general kernels."""
    Hq, Hk = heads
    b = 3
    g = torch.Generator().manual_seed(D + page + sq + sk)
    fp8_max = 448.0 if fp8_dtype == torch.float8_e4m3fn else 57344.0
    k_ref = torch.randn(b, sk, Hk, D, generator=g)
    v_ref = torch.randn(b, sk, Hk, D, generator=g)
    kd, vd = k_ref.abs().max().item() / fp8_max, v_ref.abs().max().item() / fp8_max
    kc = (k_ref / kd).to(fp8_dtype).reshape(b * sk // page, page, Hk, D)
    vc = (v_ref / vd).to(fp8_dtype).reshape(b * sk // page, page, Hk, D)
    table = torch.arange(b * sk // page, dtype=torch.int32).view(b, sk // page)
    lens = torch.tensor([sk, sk - 37, max(sq, 100)], dtype=torch.int32)
    q = torch.randn(b, sq, Hq, D, generator=g).to(torch.bfloat16)
    scale = D ** -0.5
    kf, vf = oa.dequant_fp8_cache(kc, kd), oa.dequant_fp8_cache(vc, vd)
    ks = [oa.gather_paged(kf, table[i], int(lens[i])) for i in range(b)]
    vs = [oa.gather_paged(vf, table[i], int(lens[i])) for i in range(b)]
    cu_q = torch.arange(0, b + 1, dtype=torch.int32) * sq
    ref, _ = oa.attention_ragged(q.view(-1, Hq, D), ks, vs, cu_q, scale, causal=causal)
    for layout in ("scalar", "expanded"):
        kdt = torch.tensor([kd], dtype=torch.float32, device=dev)
        vdt = torch.tensor([vd], dtype=torch.float32, device=dev)
        if layout == "expanded":
            kdt, vdt = kdt.expand(b, Hk), vdt.expand(b, Hk)
        out = sglk.flash_attn_with_kvcache(q.to(dev), kc.to(dev), vc.to(dev), cache_seqlens=lens.to(dev),
                                           page_table=table.to(dev), k_descale=kdt, v_descale=vdt, softmax_scale=scale,
                                           causal=causal)
        out = out.reshape(b * sq, Hq, D).float().cpu()
        diff = (out - ref).abs()
        # reference bounds (:1824-1830): e5m2 max 4e-1 / mean 8e-2, e4m3 max 1e-1 / mean 2e-2. The kernel converts the
        # stored fp8 values exactly and works in fp32, so it only differs from the oracle by bf16 rounding of P and out.
        assert diff.max().item() <= 2e-2 and diff.mean().item() <= 2e-3, (layout, diff.max().item(), diff.mean().item())


def test_q_descale_is_accepted_and_ignored(sglk, dev):
    """the reference parses q_descale and never reads it (q is always 16-bit there, flash_attention.cpp:290, :308-312)"""
    g = torch.Generator().manual_seed(5)
    q = torch.randn(2, 1, 8, 128, generator=g).to(torch.bfloat16).to(dev)
    kc = torch.randn(4, 64, 2, 128, generator=g).to(torch.bfloat16).to(dev)
    vc = torch.randn(4, 64, 2, 128, generator=g).to(torch.bfloat16).to(dev)
    lens = torch.tensor([100, 128], dtype=torch.int32, device=dev)
    table = torch.tensor([[0, 1], [2, 3]], dtype=torch.int32, device=dev)
    a = sglk.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, page_table=table)
    b = sglk.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, page_table=table,
                                     q_descale=torch.full((2, 2), 3.0, device=dev))
    assert torch.equal(a, b)


def test_fp8_kvcache_requires_descale(sglk, dev):
    kc = torch.zeros(2, 64, 2, 128, device=dev).to(torch.float8_e4m3fn)
    q = torch.zeros(1, 1, 2, 128, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="k_descale and v_descale"):
        sglk.flash_attn_with_kvcache(q, kc, kc, cache_seqlens=torch.tensor([5], dtype=torch.int32, device=dev),
                                     page_table=torch.zeros(1, 2, dtype=torch.int32, device=dev))


# ------------------------------------------- KV-cache addressing: 4-D cache rows, cache_batch_idx, cache_leftpad
@pytest.mark.parametrize("layout", ["rows", "paged"])
@pytest.mark.parametrize("use_batch_idx", [False, True])
@pytest.mark.parametrize("use_leftpad", [False, True])
@pytest.mark.parametrize("sq,D,heads", [(1, 128, (16, 4)), (40, 64, (8, 8)), (1, 256, (8, 1)), (130, 128, (4, 2))])
def test_kvcache_rows_batch_idx_leftpad(sglk, dev, layout, use_batch_idx, use_leftpad, sq, D, heads):
    """reference decode::mha_fwd_nopage (flash_attention.cpp:83-270), kv_batch_idx (:383, :649-653), leftpad_k (:408-412)"""
    Hq, Hk = heads
    g = torch.Generator().manual_seed(sq + D + Hq)
    b, rows, cache_len, page = 3, 5, 448, 64
    q = torch.randn(b, sq, Hq, D, generator=g).to(torch.bfloat16)
    k_rows = torch.randn(rows, cache_len, Hk, D, generator=g).to(torch.bfloat16)
    v_rows = torch.randn(rows, cache_len, Hk, D, generator=g).to(torch.bfloat16)
    batch_idx = torch.randperm(rows, generator=g)[:b].to(torch.int32) if use_batch_idx else None
    ends = torch.randint(cache_len // 2, cache_len + 1, (b,), generator=g, dtype=torch.int32)
    ends[0] = cache_len
    leftpad = None
    if use_leftpad:
        leftpad = torch.tensor([int(torch.randint(0, int(e) - sq + 1, (1,), generator=g)) for e in ends], dtype=torch.int32)
        leftpad[1] = 37  # (not a multiple of the 32-token tile)
    if layout == "rows":
        kc, vc, table = k_rows, v_rows, None
    else:  # the same cache rows cut into shuffled pages; page_table row r describes cache row r
        pages_per = cache_len // page
        perm = torch.randperm(rows * pages_per, generator=g)
        kc = torch.empty(rows * pages_per, page, Hk, D, dtype=torch.bfloat16)
        vc = torch.empty_like(kc)
        kc[perm] = k_rows.view(rows * pages_per, page, Hk, D)
        vc[perm] = v_rows.view(rows * pages_per, page, Hk, D)
        table = perm.view(rows, pages_per).to(torch.int32)
    if not use_batch_idx:  # one cache row per sequence
        kc_use, vc_use = (kc[:b], vc[:b]) if layout == "rows" else (kc, vc)
        table_use = table[:b] if table is not None else None
    else:
        kc_use, vc_use, table_use = kc, vc, table
    d = lambda t: t.to(dev) if t is not None else None
    for causal in (False, True):
        out, lse, *_ = sglk.flash_attn_with_kvcache(d(q), d(kc_use), d(vc_use), cache_seqlens=d(ends), cache_batch_idx=d(batch_idx),
                                                    cache_leftpad=d(leftpad), page_table=d(table_use), causal=causal,
                                                    return_softmax_lse=True)
        out = out.cpu().view(b, sq, Hq, D)
        for i in range(b):
            row = int(batch_idx[i]) if use_batch_idx else i
            lo, hi = (int(leftpad[i]) if use_leftpad else 0), int(ends[i])
            ki, vi = k_rows[row, lo:hi], v_rows[row, lo:hi]
            ref, ref_lse = oa.attention_seq(q[i], ki, vi, D ** -0.5, causal=causal)
            check(out[i], ref, pt_seq(q[i], ki, vi, D ** -0.5, causal, (-1, -1), 0.0, None), f"seq {i}")
            torch.testing.assert_close(lse.cpu()[:, i * sq:(i + 1) * sq], ref_lse, rtol=1e-3, atol=1e-3)


def test_kvcache_golden(sglk, dev):
    for c in load_golden("attention_kvcache"):
        out = sglk.flash_attn_with_kvcache(c["q"].to(dev), c["k_cache"].to(dev), c["v_cache"].to(dev),
                                           cache_seqlens=c["cache_seqlens"].to(dev), cache_batch_idx=c["cache_batch_idx"].to(dev),
                                           cache_leftpad=c["cache_leftpad"].to(dev), causal=c["causal"]).cpu().view(c["q"].shape)
        err = (out.float() - c["out"].float()).abs().max().item()
        err_pt = (c["out_pt"].float() - c["out"].float()).abs().max().item()
        assert err <= 2 * err_pt + 1e-5, (err, err_pt)


def test_kvcache_addressing_errors(sglk, dev):
    q = torch.zeros(2, 1, 4, 64, dtype=torch.float16, device=dev)
    kc = torch.zeros(2, 128, 4, 64, dtype=torch.float16, device=dev)
    lens = torch.tensor([5, 6], dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="int32"):
        sglk.flash_attn_with_kvcache(q, kc, kc, cache_seqlens=lens, cache_leftpad=torch.zeros(2, dtype=torch.int64, device=dev))
    with pytest.raises(RuntimeError, match="one entry per sequence"):
        sglk.flash_attn_with_kvcache(q, kc, kc, cache_seqlens=lens, cache_batch_idx=torch.zeros(3, dtype=torch.int32, device=dev))
