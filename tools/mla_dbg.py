import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib.util, os
spec = importlib.util.spec_from_file_location("sgl_kernel", "sgl-kernel-xpu_amd/python/sgl_kernel/__init__.py")
sglk = importlib.util.module_from_spec(spec); sys.modules["sgl_kernel"] = sglk; spec.loader.exec_module(sglk)
from oracle import mla as omla
dev = "cuda"
g = torch.Generator().manual_seed(42)
dtype = torch.bfloat16; bs = 2; H = 128; page = 32; seqs = [128] * bs
seq_lens = torch.tensor(seqs, dtype=torch.int32)
block_num = 4
q = (torch.randn(bs, H, 576, generator=g) * 100).to(dtype)
table = torch.randint(0, bs * block_num, (bs, block_num), generator=g, dtype=torch.int32)
cache = torch.randn(bs * block_num, page, 576, generator=g).to(dtype)
scale = (128 + 64) ** -0.5
ref = omla.mla_decode(q, cache, scale, table, seq_lens).float()
qd = q.to(dev)
ws = torch.empty(sglk.flash_mla_get_workspace_size(block_num * page, bs, H, page, num_kv_splits=1), device=dev, dtype=torch.uint8)
out = sglk.flash_mla_decode(qd[:, :, :512].contiguous(), qd[:, :, 512:].clone(), cache.to(dev), seq_lens.to(dev), table.to(dev), ws, scale, 1).cpu().float()
d = (out - ref).abs()
bad = (d > 0.01 + 0.01 * ref.abs()).sum(-1)
sl2 = scale * 1.4426950408889634
for b in range(bs):
    toks = torch.arange(seqs[b])
    rows = cache[table[b, toks // page].long(), toks % page].float()
    S = (q[b].float() @ rows.T) * sl2
    for h in range(H):
        if bad[b, h] > 0:
            mt = S[h].view(-1, 32).max(1).values
            top = S[h].topk(3)
            print("b", b, "h", h, "bad", int(bad[b, h]), "maxerr %.4f" % d[b, h].max().item(), "tile max", [round(x, 1) for x in mt.tolist()],
                  "top3", [round(x, 1) for x in top.values.tolist()], top.indices.tolist())
import ctypes
lib = ctypes.CDLL("sgl-kernel-xpu_amd/python/sgl_kernel/libsglk.so")
def run(w):
    lib.sglk_debug_set_mla_waves_per_group(w)
    return sglk.flash_mla_decode(qd[:, :, :512].contiguous(), qd[:, :, 512:].clone(), cache.to(dev), seq_lens.to(dev), table.to(dev), ws, scale, 1).cpu().float()
o_new, o_old = run(0), run(1)
for (b, h) in [(0, 43), (0, 105), (1, 1), (1, 44)]:
    toks = torch.arange(seqs[b])
    rows = cache[table[b, toks // page].long(), toks % page].float()
    S = (q[b].float() @ rows.T) * sl2
    top = S[h].topk(2)
    V1, V2 = rows[top.indices[0], :512], rows[top.indices[1], :512]
    for name, o in (("ref", ref), ("old", o_old), ("new", o_new)):
        w2 = ((o[b, h] - V1) @ (V2 - V1)) / ((V2 - V1) @ (V2 - V1))
        print(b, h, name, "w2 %.4f" % w2.item(), "log2(w2/w1) %.3f" % torch.log2(w2 / (1 - w2)).item(), "true %.3f" % (top.values[1] - top.values[0]).item())
