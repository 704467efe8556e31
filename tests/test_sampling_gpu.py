"""GPU parity: top_k_renorm_probs, top_p_renorm_probs, top_k_top_p_sampling_from_probs, top_p_sampling_from_probs,
min_p_sampling_from_probs vs the CPU oracle's filters.

Grids follow reference tests/test_sampling.py (batch {1, 99, 989}, vocab {111, 32000, 128256, 151936}, scalar and
per-row thresholds); draws are checked the way the reference does (every sample lies inside the filter mask, :56-61),
plus what an exact-pivot implementation can promise on top: the mask of renormalised rows EQUALS the oracle's,
draws are reproducible from the generator state, and the empirical distribution matches."""
import pytest
import torch
from conftest import load_golden

from oracle import sampling as osamp

pytestmark = pytest.mark.gpu


def make_probs(B, V, seed=42):
    g = torch.Generator().manual_seed(seed)
    pre = torch.rand(B, V, generator=g)
    return pre / pre.sum(dim=-1, keepdim=True)


@pytest.mark.parametrize("B", [1, 99, 989])
@pytest.mark.parametrize("V", [111, 32000, 128256])
@pytest.mark.parametrize("k", [10, 100, 500, "array"])
def test_top_k_renorm(sglk, dev, B, V, k):
    if k != "array" and k > V:
        pytest.skip("k > vocab")
    if B == 989 and V > 32000:
        B = 123  # (CPU oracle time)
    pr = make_probs(B, V)
    kk = torch.randint(10, min(200, V), (B,), generator=torch.Generator().manual_seed(1)) if k == "array" else k
    out = sglk.top_k_renorm_prob(pr.to(dev), kk.to(dev) if k == "array" else kk).cpu()
    mask = osamp.top_k_mask(pr, kk)
    assert torch.equal(out > 0, mask & (pr > 0))  # exact pivot: the kept set is the oracle's
    torch.testing.assert_close(out, osamp.renorm(pr, mask), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.sum(dim=-1), torch.ones(B), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B", [1, 99, 989])
@pytest.mark.parametrize("V", [111, 32000, 151936])
@pytest.mark.parametrize("p", [0.1, 0.5, 1.0, "array"])
def test_top_p_renorm(sglk, dev, B, V, p):
    if B == 989 and V > 32000:
        B = 123
    pr = make_probs(B, V)
    pp = torch.rand(B, generator=torch.Generator().manual_seed(2)) * 0.8 + 0.1 if p == "array" else p
    out = sglk.top_p_renorm_prob(pr.to(dev), pp.to(dev) if p == "array" else pp).cpu()
    mask = osamp.top_p_mask(pr, pp)
    got = out > 0
    # the nucleus boundary is decided on sums of ~V terms: the oracle sums in float64, the kernel in 2^-40 fixed point;
    # at most the boundary element of a row may differ
    assert ((got != (mask & (pr > 0))).sum(dim=1) <= 1).all()
    torch.testing.assert_close(out, osamp.renorm(pr, got), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("B", [1, 99, 989])
@pytest.mark.parametrize("V", [111, 32000, 128256])
@pytest.mark.parametrize("p,kfrac", [(0.1, 0.5), (0.5, 0.1), ("array", None)])
def test_joint_sampling_membership(sglk, dev, B, V, p, kfrac):
    if B == 989 and V > 32000:
        B = 123
    pr = make_probs(B, V)
    if p == "array":
        g = torch.Generator().manual_seed(3)
        kk = torch.randint(10, min(200, V), (B,), generator=g)
        pp = torch.rand(B, generator=g) * 0.6 + 0.1
        k_dev, p_dev = kk.to(dev), pp.to(dev)
    else:
        kk, pp = int(V * kfrac), p
        k_dev, p_dev = kk, pp
    mask = osamp.top_k_mask(pr, kk) & osamp.top_p_mask(pr, pp)
    # one element of slack at the nucleus boundary (see test_top_p_renorm): allow the kernel's own renorm mask as well
    mask |= sglk.top_p_renorm_prob(pr.to(dev), p_dev).cpu() > 0 if p == "array" else mask
    mask &= osamp.top_k_mask(pr, kk)
    prd = pr.to(dev)
    rows = torch.arange(B)
    gen = torch.Generator(device=dev).manual_seed(7)
    for _ in range(50):
        s = sglk.top_k_top_p_sampling_from_probs(prd, k_dev, p_dev, filter_apply_order="joint", generator=gen).cpu().long()
        assert s.dtype == torch.int64 and torch.all((s >= 0) & (s < V))
        assert torch.all(mask[rows, s]), pr[rows, s][~mask[rows, s]]


@pytest.mark.parametrize("B", [1, 99])
@pytest.mark.parametrize("V", [111, 32000, 128256])
@pytest.mark.parametrize("p", [0.05, 0.2, 0.7, 1.0, "array"])
def test_min_p_sampling_membership(sglk, dev, B, V, p):
    pr = make_probs(B, V)
    pp = torch.rand(B, generator=torch.Generator().manual_seed(4)) * 0.6 + 0.05 if p == "array" else p
    mask = osamp.min_p_mask(pr, pp)
    prd, p_dev = pr.to(dev), (pp.to(dev) if p == "array" else pp)
    rows = torch.arange(B)
    for _ in range(50):
        s = sglk.min_p_sampling_from_probs(prd, p_dev).cpu().long()
        assert torch.all(mask[rows, s])


def test_draws_are_reproducible_and_distributed_like_the_filtered_probs(sglk, dev):
    # 20000 rows of ONE small distribution: the sample histogram must follow the renormalised kept probabilities
    V, B = 16, 20000
    base = torch.tensor([0.30, 0.02, 0.18, 0.01, 0.09, 0.05, 0.11, 0.004, 0.07, 0.03, 0.06, 0.016, 0.02, 0.02, 0.01, 0.01])
    base = base / base.sum()
    pr = base.unsqueeze(0).repeat(B, 1).contiguous().to(dev)
    g1 = torch.Generator(device=dev).manual_seed(123)
    g2 = torch.Generator(device=dev).manual_seed(123)
    a = sglk.top_k_top_p_sampling_from_probs(pr, 8, 0.85, filter_apply_order="joint", generator=g1)
    b = sglk.top_k_top_p_sampling_from_probs(pr, 8, 0.85, filter_apply_order="joint", generator=g2)
    assert torch.equal(a, b)  # same generator state, same draws
    c = sglk.top_k_top_p_sampling_from_probs(pr, 8, 0.85, filter_apply_order="joint", generator=g1)
    assert not torch.equal(a, c)  # the generator advanced
    mask = (osamp.top_k_mask(base[None], 8) & osamp.top_p_mask(base[None], 0.85))[0]
    want = osamp.renorm(base[None], mask[None])[0]
    freq = torch.bincount(a.cpu().long(), minlength=V).float() / B
    assert torch.all(freq[~mask] == 0)
    assert (freq - want).abs().max() < 4 * (want * (1 - want) / B).sqrt().max() + 1e-3  # 4 sigma
    # indices: output row b samples probs row indices[b]
    two = torch.stack([base, torch.eye(V)[5]]).to(dev)
    idx = torch.tensor([1, 1, 0, 1], device=dev)
    s = sglk.top_p_sampling_from_probs(two, 1.0, indices=idx).cpu()
    assert s[0] == 5 and s[1] == 5 and s[3] == 5
    # top_k_first order = renorm then top-p sampling
    s = sglk.top_k_top_p_sampling_from_probs(pr, 1, 0.99)
    assert torch.all(s.cpu() == 0)
    with pytest.raises(ValueError, match="NaN"):
        sglk.min_p_sampling_from_probs(torch.full((1, 8), float("nan"), device=dev), 0.1, check_nan=True)
    with pytest.raises(ValueError, match="Invalid filter_apply_order"):
        sglk.top_k_top_p_sampling_from_probs(pr, 1, 0.5, filter_apply_order="p_first")


def test_draw_is_the_inverse_cdf_in_row_order(sglk, dev):
    # The draw of row b is the first index at which the running (fixed-point) mass exceeds u_b * Z, u_b = Philox(seed; offset, b).
    # u_b is recovered to 17 bits from a draw over 2^17 equal probabilities at the same generator state; the index drawn from any other
    # row of probabilities must then own a CDF interval that meets [u_lo, u_hi). Rows start at every 16-byte misalignment (V odd) and
    # the longest one is cut into two-step runs (more than 262141 elements).
    B, Vu = 64, 1 << 17
    uni = torch.full((B, Vu), 1.0 / Vu, device=dev)
    g = torch.Generator(device=dev).manual_seed(99)
    iu = sglk.top_p_sampling_from_probs(uni, 1.0, generator=g).cpu().double()
    u_lo, u_hi = iu / Vu, (iu + 1) / Vu
    assert iu.unique().numel() > B // 2
    rows = torch.arange(B)
    for V in (111, 4099, 128256, 300001):
        pr = make_probs(B, V, seed=5)
        g = torch.Generator(device=dev).manual_seed(99)
        s = sglk.top_p_sampling_from_probs(pr.to(dev), 1.0, generator=g).cpu().long()
        fix = (pr.double() * 2.0 ** 40).floor()
        cdf = fix.cumsum(-1)
        Z = cdf[:, -1]
        hi = cdf[rows, s] / Z
        lo = (cdf[rows, s] - fix[rows, s]) / Z
        assert torch.all(hi > u_lo - 1e-9) and torch.all(lo < u_hi + 1e-9), V
        # min-p and joint draws share the stream: with everything kept they are the same draws
        g = torch.Generator(device=dev).manual_seed(99)
        assert torch.equal(sglk.min_p_sampling_from_probs(pr.to(dev), 0.0, generator=g).cpu().long(), s)


def test_rows_longer_than_262141_elements(sglk, dev):
    B, V = 3, 300001
    pr = make_probs(B, V, seed=11)
    out = sglk.top_k_renorm_prob(pr.to(dev), 1000).cpu()
    mask = osamp.top_k_mask(pr, 1000)
    assert torch.equal(out > 0, mask & (pr > 0))
    torch.testing.assert_close(out, osamp.renorm(pr, mask), rtol=1e-3, atol=1e-3)
    out = sglk.top_p_renorm_prob(pr.to(dev), 0.3).cpu()
    torch.testing.assert_close(out.sum(dim=-1), torch.ones(B), rtol=1e-5, atol=1e-5)
    keep = out > 0
    assert torch.all((pr * keep).sum(-1) >= 0.3 - 1e-4)
    jm = osamp.top_k_mask(pr, 5000) & keep
    gen = torch.Generator(device=dev).manual_seed(3)
    for _ in range(30):
        smp = sglk.top_k_top_p_sampling_from_probs(pr.to(dev), 5000, 0.3, filter_apply_order="joint", generator=gen).cpu().long()
        assert torch.all(jm[torch.arange(B), smp])


@pytest.mark.parametrize("B", [1, 5, 32, 64])
@pytest.mark.parametrize("V", [32768, 128256, 151936])
def test_cluster_form_returns_the_bits_of_the_one_workgroup_form(sglk, dev, B, V):
    # up to 64 rows of >= 32768 entries run as several workgroups per row with a scratch tensor (sglk_sampling_ws); the same rows inside
    # a batch of 80 run one workgroup per row: equal renormalised rows, equal draws from equal generator states - eager and in a graph
    big = make_probs(80, V, seed=B + V).to(dev)
    small = big[:B].contiguous()
    kk = torch.randint(5, 400, (80,), generator=torch.Generator().manual_seed(1)).to(dev)
    pp = (torch.rand(80, generator=torch.Generator().manual_seed(2)) * 0.8 + 0.1).to(dev)
    assert torch.equal(sglk.top_k_renorm_prob(small, kk[:B]), sglk.top_k_renorm_prob(big, kk)[:B])
    assert torch.equal(sglk.top_p_renorm_prob(small, pp[:B]), sglk.top_p_renorm_prob(big, pp)[:B])
    assert torch.equal(sglk.top_p_renorm_prob(small, 1.0), sglk.top_p_renorm_prob(big, 1.0)[:B])
    calls = (lambda pr, n, g: sglk.top_k_top_p_sampling_from_probs(pr, kk[:n].int(), pp[:n], filter_apply_order="joint", generator=g),
             lambda pr, n, g: sglk.top_k_top_p_sampling_from_probs(pr, 20, 0.7, generator=g),
             lambda pr, n, g: sglk.top_p_sampling_from_probs(pr, pp[:n], generator=g),
             lambda pr, n, g: sglk.top_p_sampling_from_probs(pr, 1.0, generator=g),
             lambda pr, n, g: sglk.min_p_sampling_from_probs(pr, pp[:n] * 0.05, generator=g),
             lambda pr, n, g: sglk.min_p_sampling_from_probs(pr, 0.0, generator=g))
    for f in calls:
        for rep in range(3):
            g1 = torch.Generator(device=dev).manual_seed(77 + rep)
            g2 = torch.Generator(device=dev).manual_seed(77 + rep)
            assert torch.equal(f(small, B, g1), f(big, 80, g2)[:B])
    # all-zero rows draw index 0 in both forms
    z = torch.zeros(B, V, device=dev)
    assert torch.all(sglk.top_p_sampling_from_probs(z, 0.5) == 0)
    # recorded into a graph: replays follow the default generator as the eager calls do
    torch.cuda.manual_seed(11)
    eager = [sglk.top_k_top_p_sampling_from_probs(small, 50, 0.9, filter_apply_order="joint").clone() for _ in range(2)]
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = sglk.top_k_top_p_sampling_from_probs(small, 50, 0.9, filter_apply_order="joint")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.manual_seed(11)
    for want in eager:
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


def test_draws_recorded_into_a_graph_follow_the_generator(sglk, dev):
    # A sampling launch recorded into a HIP graph reads the generator state from the device at replay (the C-ABI's *_graph entries):
    # every replay draws fresh numbers, and they are the numbers the eager op draws from the same generator state.
    V, B = 4096, 256
    pr = make_probs(B, V, seed=7).to(dev)
    torch.cuda.manual_seed(2024)
    eager = [(sglk.top_k_top_p_sampling_from_probs(pr, 50, 0.9, filter_apply_order="joint"), sglk.top_p_sampling_from_probs(pr, 0.8),
              sglk.min_p_sampling_from_probs(pr, 0.01)) for _ in range(3)]
    assert not torch.equal(eager[0][0], eager[1][0])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            a = sglk.top_k_top_p_sampling_from_probs(pr, 50, 0.9, filter_apply_order="joint")
            b = sglk.top_p_sampling_from_probs(pr, 0.8)
            c = sglk.min_p_sampling_from_probs(pr, 0.01)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.manual_seed(2024)
    for want in eager:
        graph.replay()
        torch.cuda.synchronize()
        for got, w in zip((a, b, c), want):
            assert torch.equal(got, w)


def test_sampling_golden(sglk, dev):
    gold = load_golden("sampling")
    mv = lambda v: v.to(dev) if isinstance(v, torch.Tensor) else v
    for c in gold["top_k_renorm"]:
        torch.testing.assert_close(sglk.top_k_renorm_prob(c["probs"].to(dev), mv(c["k"])).cpu(), c["out"], rtol=1e-3, atol=1e-3)
    for c in gold["top_p_renorm"]:
        torch.testing.assert_close(sglk.top_p_renorm_prob(c["probs"].to(dev), mv(c["p"])).cpu(), c["out"], rtol=1e-3, atol=1e-3)
    for c in gold["joint_mask"]:
        B = c["probs"].shape[0]
        for _ in range(20):
            s = sglk.top_k_top_p_sampling_from_probs(c["probs"].to(dev), mv(c["k"]), mv(c["p"]), filter_apply_order="joint").cpu().long()
            assert torch.all(c["mask"][torch.arange(B), s] == 1)
    for c in gold["min_p_mask"]:
        B = c["probs"].shape[0]
        for _ in range(20):
            s = sglk.min_p_sampling_from_probs(c["probs"].to(dev), mv(c["p"])).cpu().long()
            assert torch.all(c["mask"][torch.arange(B), s] == 1)
