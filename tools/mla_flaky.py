"""Repeat flash_mla_decode on one input and compare runs with each other (race detector)."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
lib = ctypes.CDLL(os.path.join(os.path.dirname(sgl_kernel.__file__), "libsglk.so"))
dev = "cuda"
for H in (16, 32, 64, 128):
    for page in (32, 64):
        for seq in (128, 1024):
            for forced in (0, 1):
                if forced and H == 128:
                    continue
                lib.sglk_debug_set_mla_waves_per_group(forced)
                torch.manual_seed(H + page)
                bs = 4
                nblk = (seq + page - 1) // page
                q = (torch.randn(bs, H, 576, device=dev) * 100).to(torch.bfloat16)
                cache = torch.randn(bs * nblk, page, 576, device=dev).to(torch.bfloat16)
                table = torch.randint(0, bs * nblk, (bs, nblk), device=dev, dtype=torch.int32)
                lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
                qn, qp = q[..., :512].contiguous(), q[..., 512:].contiguous()
                ws = torch.empty(sgl_kernel.flash_mla_get_workspace_size(seq, bs, H, page, 1), device=dev, dtype=torch.uint8)
                ref = None
                bad = 0
                where = set()
                for it in range(200):
                    o = sgl_kernel.flash_mla_decode(qn, qp, cache, lens, table, ws, 0.072, 1).float()
                    if ref is None:
                        ref = o.clone()
                    elif not torch.equal(o, ref):
                        bad += 1
                        idx = (o != ref).nonzero()
                        for b_, h_, d_ in idx[:64].tolist():
                            where.add((h_, d_ // 16))
                print(f"H={H} page={page} seq={seq} forcedW={forced}: {bad}/199 runs differ", sorted(where)[:12])
