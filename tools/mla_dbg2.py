import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("sgl_kernel", "sgl-kernel-xpu_amd/python/sgl_kernel/__init__.py")
sglk = importlib.util.module_from_spec(spec); sys.modules["sgl_kernel"] = sglk; spec.loader.exec_module(sglk)
from oracle import mla as omla
dev = "cuda"
import itertools
for seqlen, QS in itertools.product((20, 64, 200), (3.0, 10.0, 30.0, 100.0)):
    g = torch.Generator().manual_seed(42)
    dtype = torch.bfloat16; bs = 2; H = 128; page = 32; seqs = [seqlen] * bs
    seq_lens = torch.tensor(seqs, dtype=torch.int32)
    block_num = 8
    q = (torch.randn(bs, H, 576, generator=g) * QS).to(dtype)
    table = torch.randint(0, bs * block_num, (bs, block_num), generator=g, dtype=torch.int32)
    cache = torch.randn(bs * block_num, page, 576, generator=g).to(dtype)
    scale = (128 + 64) ** -0.5
    ref = omla.mla_decode(q, cache, scale, table, seq_lens).float()
    qd = q.to(dev)
    ws = torch.empty(max(1, sglk.flash_mla_get_workspace_size(block_num * page, bs, H, page, num_kv_splits=1)), device=dev, dtype=torch.uint8)
    out = sglk.flash_mla_decode(qd[:, :, :512].contiguous(), qd[:, :, 512:].clone(), cache.to(dev), seq_lens.to(dev), table.to(dev), ws, scale, 1).cpu().float()
    nan = torch.isnan(out)
    print("seq", seqlen, "qs", QS, "nan frac", nan.float().mean().item(), "nan heads", nan.any(-1).sum().item(), "nan dims(b0,h0)", nan[0, 0].sum().item(),
          "maxerr(non-nan)", (out - ref)[~nan].abs().max().item() if (~nan).any() else None)
    if nan.any():
        hh = nan.any(-1)[0].nonzero().flatten().tolist()
        print("  b0 nan heads:", hh[:40])
        print("  dims nan for first nan head:", nan[0, hh[0]].nonzero().flatten().tolist()[:40] if hh else None)
