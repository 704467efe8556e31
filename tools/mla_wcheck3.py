import os, sys, ctypes
import torch
root = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(root, "sgl-kernel-xpu_amd", "python")); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import sgl_kernel
from oracle import mla as omla
lib = ctypes.CDLL(os.path.join(os.path.dirname(sgl_kernel.__file__), "libsglk.so"))
dev = "cuda"
def case(dtype, seqs, page, H, splits, seed=42, q_scale=100.0):
    g = torch.Generator().manual_seed(seed)
    bs = len(seqs)
    seq_lens = torch.tensor(seqs, dtype=torch.int32)
    block_num = (max(max(seqs), 1) + page - 1) // page
    pack = 128 // page
    block_num = (block_num + pack - 1) // pack * pack
    q = (torch.randn(bs, H, 576, generator=g) * q_scale).to(dtype)
    table = torch.randint(0, bs * block_num, (bs, block_num), generator=g, dtype=torch.int32)
    cache = torch.randn(bs * block_num, page, 576, generator=g).to(dtype)
    scale = (128 + 64) ** -0.5
    ref = omla.mla_decode(q, cache, scale, table, seq_lens)
    qd = q.to(dev)
    qn, qp = qd[:, :, :512].contiguous(), qd[:, :, 512:].clone()
    for forced in (4, 2, 1):
        lib.sglk_debug_set_mla_waves_per_group(forced)
        ws = torch.empty(sgl_kernel.flash_mla_get_workspace_size(block_num * page, bs, H, page, splits), device=dev, dtype=torch.uint8)
        res = []
        for it in range(3):
            out = sgl_kernel.flash_mla_decode(qn, qp, cache.to(dev), seq_lens.to(dev), table.to(dev), ws, scale, splits)
            err = (out.cpu().float() - ref.float()).abs()
            bad = (err > 1e-2 + 1e-2 * ref.float().abs()).nonzero()
            res.append((round(err.max().item(), 4), len(bad), sorted({(b_, h_, d_ // 16) for b_, h_, d_ in bad[:2000].tolist()})[:8]))
        print(dtype, seqs, page, H, splits, "W", forced, res)
case(torch.float16, [513, 2000], 128, 32, 1, q_scale=3.0)
case(torch.bfloat16, [513, 2000], 128, 32, 1, q_scale=3.0)
case(torch.float16, [513, 2000], 128, 32, 1, q_scale=3.0, seed=7)
case(torch.float16, [1000, 500, 1007], 64, 32, 1, seed=1000)
