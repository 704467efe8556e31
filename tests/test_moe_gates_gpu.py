"""GPU parity: topk_sigmoid, biased_topk, moe_fused_gate vs the CPU oracle. Index paths are exact:
the selected expert ids must EQUAL the oracle's (same iterative arg-max, ties -> lower index), except where two
candidates' fp32 scores differ by the last bit between the device's expf and the host's (then the swapped experts must
have equal choice values up to 1 ulp: the reference tests' own acceptance, tests/test_topk_sigmoid.py:13-38).
Grids follow reference tests/test_topk_sigmoid.py:101-110, tests/test_biased_topk.py:88-95,
tests/test_moe_fused_gate.py:141-160."""
import pytest
import torch
from conftest import load_golden

from oracle import moe_gates as og

pytestmark = pytest.mark.gpu


def check_routing(choice, w, ids, ref_w, ref_ids, n_cols, rtol=1e-4, atol=1e-5):
    """ids equal as sets per row, or differing only between candidates whose choice scores tie within 2 ulp."""
    ids_c, ref_c = ids.cpu().long(), ref_ids.long()
    same = ids_c.sort(dim=1).values == ref_c.sort(dim=1).values
    bad_rows = (~same.all(dim=1)).nonzero().flatten().tolist()
    for r in bad_rows:
        ours, theirs = set(ids_c[r].tolist()), set(ref_c[r].tolist())
        more = sorted(choice[r, i].item() for i in ours - theirs if i < choice.shape[1])
        less = sorted(choice[r, i].item() for i in theirs - ours if i < choice.shape[1])
        assert len(more) == len(less) and all(abs(a - b) <= 4e-7 * max(1.0, abs(a)) for a, b in zip(more, less)), (r, more, less)
    assert len(bad_rows) <= max(1, ids.shape[0] // 200), f"{len(bad_rows)} rows differ"
    ok = torch.ones(ids.shape[0], dtype=torch.bool)
    ok[bad_rows] = False
    dense = torch.zeros(ids.shape[0], n_cols).scatter_(1, ids_c, w.cpu().float())
    ref = torch.zeros(ids.shape[0], n_cols).scatter_(1, ref_c, ref_w.float())
    torch.testing.assert_close(dense[ok], ref[ok], rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("T", [2, 32, 4096])
@pytest.mark.parametrize("E", [8, 32, 256, 384])
@pytest.mark.parametrize("renorm", [False, True])
@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("topk,shared", [(1, 0), (2, 0), (2, 1), (4, 0), (4, 1), (8, 2)])
def test_topk_sigmoid(sglk, dev, dtype, T, E, renorm, with_bias, topk, shared):
    g = torch.Generator().manual_seed(1024 + E + T)
    x = torch.randn(T, E, generator=g).to(dtype)
    bias = torch.randn(E, generator=g) if with_bias else None
    rsf = 2.5 if (T + E) % 2 else 1.0
    w = torch.empty(T, topk, dtype=torch.float32, device=dev)
    ids = torch.empty(T, topk, dtype=torch.int32, device=dev)
    sglk.topk_sigmoid(w, ids, x.to(dev), renorm, bias.to(dev) if with_bias else None, rsf, shared)
    rw, rids = og.topk_sigmoid(x, topk, renorm, bias, rsf, shared)
    choice = torch.sigmoid(x.float()) + (bias if with_bias else 0.0)
    check_routing(choice, w, ids, rw, rids, E + max(shared, 1))


@pytest.mark.parametrize("T", [1, 64, 1024])
@pytest.mark.parametrize("E", [128, 384, 512])
@pytest.mark.parametrize("topk", [4, 6, 8])
@pytest.mark.parametrize("scoring", ["sigmoid", "sqrtsoftplus"])
@pytest.mark.parametrize("shared", [0, 1])
@pytest.mark.parametrize("renorm,apply", [(True, True), (True, False), (False, True), (False, False)])
def test_biased_topk(sglk, dev, T, E, topk, scoring, shared, renorm, apply):
    g = torch.Generator().manual_seed(E * 100 + topk)
    x = torch.randn(T, E, generator=g) * 2.0
    bias = torch.randn(E, generator=g) * 0.5
    w = torch.empty(T, topk, dtype=torch.float32, device=dev)
    ids = torch.empty(T, topk, dtype=torch.int32, device=dev)
    sglk.biased_topk(x.to(dev), bias.to(dev), w, ids, topk, scoring, shared, renorm, 2.5, apply)
    rw, rids = og.biased_topk(x, bias, topk, scoring, shared, renorm, 2.5, apply)
    check_routing(og._score(x, scoring) + bias, w, ids, rw, rids, E + max(shared, 1))


@pytest.mark.parametrize("T", [1, 2, 7, 64, 1024, 16384])
@pytest.mark.parametrize("E,G,tg,topk", [(128, 4, 2, 4), (256, 8, 4, 8), (512, 16, 8, 16), (256, 16, 4, 6), (64, 1, 1, 6)])
@pytest.mark.parametrize("scoring", ["sigmoid", "softmax"])
@pytest.mark.parametrize("renorm,apply", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("shared", [0, 2])
def test_moe_fused_gate(sglk, dev, T, E, G, tg, topk, scoring, renorm, apply, shared):
    g = torch.Generator().manual_seed(T)
    x = torch.rand(T, E, generator=g)
    bias = None if scoring == "softmax" else torch.rand(E, generator=g)
    w, ids = sglk.moe_fused_gate(x.to(dev), bias.to(dev) if bias is not None else None, G, tg, topk + shared, renormalize=renorm,
                                 scoring_func=scoring, num_fused_shared_experts=shared, routed_scaling_factor=2.5,
                                 apply_routed_scaling_factor_on_output=apply)
    assert w.shape == (T, topk + shared) and w.dtype == torch.float32 and ids.dtype == torch.int32
    rw, rids = og.moe_fused_gate(x, bias, G, tg, topk + shared, shared, scoring, renorm, 2.5, apply)
    choice = og._score(x, scoring) + (bias if bias is not None else 0.0)
    check_routing(choice, w, ids, rw, rids, E + max(shared, 1), rtol=1e-4, atol=1e-6)


def test_moe_fused_gate_bf16_and_exact_ties(sglk, dev):
    # exact ties everywhere: the lower index must win, in the group ranking and in the expert ranking
    x = torch.zeros(3, 256, dtype=torch.bfloat16)
    w, ids = sglk.moe_fused_gate(x.to(dev), torch.zeros(256, dtype=torch.bfloat16, device=dev), 8, 4, 8)
    assert ids.cpu().tolist() == [list(range(8))] * 3
    assert torch.allclose(w.cpu(), torch.full((3, 8), 0.125))
    w = torch.empty(2, 4, device=dev)
    ids = torch.empty(2, 4, dtype=torch.int32, device=dev)
    sglk.topk_sigmoid(w, ids, torch.zeros(2, 64, dtype=torch.float16, device=dev), False)
    assert ids.cpu().tolist() == [[0, 1, 2, 3]] * 2


def test_gates_golden(sglk, dev):
    gold = load_golden("moe_gates")
    d = lambda t: t.to(dev) if t is not None else None
    for c in gold["topk_sigmoid"]:
        T = c["x"].shape[0]
        w, ids = torch.empty(T, c["topk"], device=dev), torch.empty(T, c["topk"], dtype=torch.int32, device=dev)
        sglk.topk_sigmoid(w, ids, d(c["x"]), c["renormalize"], d(c["bias"]), c["rsf"], c["shared"])
        check_routing(torch.sigmoid(c["x"]) + (c["bias"] if c["bias"] is not None else 0.0), w, ids, c["weights"], c["ids"],
                      c["x"].shape[1] + 1)
    for c in gold["biased_topk"]:
        T = c["x"].shape[0]
        w, ids = torch.empty(T, c["topk"], device=dev), torch.empty(T, c["topk"], dtype=torch.int32, device=dev)
        sglk.biased_topk(d(c["x"]), d(c["bias"]), w, ids, c["topk"], c["scoring"], c["shared"], c["renormalize"], c["rsf"], c["apply"])
        check_routing(og._score(c["x"], c["scoring"]) + c["bias"], w, ids, c["weights"], c["ids"], c["x"].shape[1] + 1)
    for c in gold["moe_fused_gate"]:
        w, ids = sglk.moe_fused_gate(d(c["x"]), d(c["bias"]), c["G"], c["topk_group"], c["topk"], renormalize=c["renormalize"],
                                     scoring_func=c["scoring"], routed_scaling_factor=c["rsf"],
                                     apply_routed_scaling_factor_on_output=c["apply"])
        choice = og._score(c["x"], c["scoring"]) + (c["bias"] if c["bias"] is not None else 0.0)
        check_routing(choice, w, ids, c["weights"], c["ids"], c["x"].shape[1], rtol=1e-2, atol=1e-3)


def test_gate_errors(sglk, dev):
    x = torch.zeros(2, 600, device=dev)
    with pytest.raises(RuntimeError, match="num_experts"):
        sglk.topk_sigmoid(torch.empty(2, 2, device=dev), torch.empty(2, 2, dtype=torch.int32, device=dev), x, False)
    with pytest.raises(ValueError, match="Unknown scoring_func"):
        sglk.biased_topk(x[:, :128], torch.zeros(128, device=dev), torch.empty(2, 2, device=dev),
                         torch.empty(2, 2, dtype=torch.int32, device=dev), 2, "tanh")
    with pytest.raises(RuntimeError, match="divisible"):
        sglk.moe_fused_gate(x[:, :100].contiguous(), None, 8, 4, 4)
