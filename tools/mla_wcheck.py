"""flash_mla_decode vs a torch fp32 reference on the GPU, for every allowed waves-per-group setting."""
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
for H, q_scale in ((16, 1.0), (32, 1.0), (32, 3.0), (64, 1.0), (32, 100.0)):
    for forced in (8, 4, 2, 1):
        if forced * ((H + 15) // 16) > 8:
            continue
        lib.sglk_debug_set_mla_waves_per_group(forced)
        torch.manual_seed(H)
        bs, page, seq = 2, 64, 1000
        nblk = (seq + page - 1) // page
        q = (torch.randn(bs, H, 576, device=dev) * q_scale).to(torch.bfloat16)
        cache = torch.randn(bs * nblk, page, 576, device=dev).to(torch.bfloat16)
        table = torch.randint(0, bs * nblk, (bs, nblk), device=dev, dtype=torch.int32)
        lens = torch.tensor([seq, 513], device=dev, dtype=torch.int32)
        qn, qp = q[..., :512].contiguous(), q[..., 512:].contiguous()
        ws = torch.empty(sgl_kernel.flash_mla_get_workspace_size(seq, bs, H, page, 1), device=dev, dtype=torch.uint8)
        o = sgl_kernel.flash_mla_decode(qn, qp, cache, lens, table, ws, 576 ** -0.5, 1).float()
        r = ref(q, cache, table, lens, 576 ** -0.5)
        err = (o - r).abs()
        bad = (err > 1e-2 + 1e-2 * r.abs()).nonzero()
        print(f"H={H} q_scale={q_scale} W={forced}: max err {err.max().item():.4f} bad {len(bad)}",
              sorted({(b_, h_, d_ // 16) for b_, h_, d_ in bad[:4000].tolist()})[:10])
