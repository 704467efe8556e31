import os, sys, ctypes
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
lib = ctypes.CDLL(os.path.join(os.path.dirname(sgl_kernel.__file__), "libsglk.so"))
dev = "cuda"
def ref(q, cache, table, lens, scale):
    bs, H, _ = q.shape
    out = torch.zeros(bs, H, 512, device=dev)
    for b in range(bs):
        kv = cache[table[b].long()].reshape(-1, 576)[: lens[b]].float()
        s = (q[b].float() @ kv.T) * scale
        out[b] = torch.softmax(s, -1) @ kv[:, :512]
    return out
for dtype in (torch.float16, torch.bfloat16):
  for page in (128, 64):
    for seqs in ([513, 2000], [1024]):
      for forced in (4, 2, 1):
        lib.sglk_debug_set_mla_waves_per_group(forced)
        H = 32
        torch.manual_seed(42)
        bs = len(seqs)
        nblk = (max(seqs) + page - 1) // page
        q = (torch.randn(bs, H, 576, device=dev) * 3).to(dtype)
        cache = torch.randn(bs * nblk, page, 576, device=dev).to(dtype)
        table = torch.randint(0, bs * nblk, (bs, nblk), device=dev, dtype=torch.int32)
        lens = torch.tensor(seqs, device=dev, dtype=torch.int32)
        qn, qp = q[..., :512].contiguous(), q[..., 512:].contiguous()
        ws = torch.empty(16, device=dev, dtype=torch.uint8)
        r = ref(q, cache, table, lens, 192 ** -0.5)
        res = []
        for it in range(5):
            o = sgl_kernel.flash_mla_decode(qn, qp, cache, lens, table, ws, 192 ** -0.5, 1).float()
            err = (o - r).abs()
            bad = (err > 2e-3 + 2e-3 * r.abs()).nonzero()
            res.append((round(err.max().item(), 4), len(bad), sorted({(h_, d_ // 16) for b_, h_, d_ in bad[:4000].tolist()})[:6]))
        print(f"{dtype} page={page} seqs={seqs} W={forced}:", res)
