import os, sys, ctypes
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
lib = ctypes.CDLL(os.path.join(os.path.dirname(sgl_kernel.__file__), "libsglk.so"))
dev = "cuda"
H, page, seq, bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 64, 1024, 2
torch.manual_seed(42)
nblk = seq // page
q = (torch.randn(bs, H, 576, device=dev) * 3).to(torch.bfloat16)
cache = torch.randn(bs * nblk, page, 576, device=dev).to(torch.bfloat16)
table = torch.randint(0, bs * nblk, (bs, nblk), device=dev, dtype=torch.int32)
lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
qn, qp = q[..., :512].contiguous(), q[..., 512:].contiguous()
ws = torch.empty(16, device=dev, dtype=torch.uint8)
run = lambda: sgl_kernel.flash_mla_decode(qn, qp, cache, lens, table, ws, 192 ** -0.5, 1).float()
lib.sglk_debug_set_mla_probe(0)
run(); ref = run()
for point in range(1, 8):
    res = []
    for wave in range(8):
        lib.sglk_debug_set_mla_probe(16 * point + wave)
        o = run()
        res.append(int((o != ref).sum()))
    print(f"H={H} point {point}: differing elements per stalled wave:", res)
for point, wave in ((4, 4), (4, 0), (4, 5)):
    lib.sglk_debug_set_mla_probe(16 * point + wave)
    for it in range(3):
        o = run()
        d = (o != ref).nonzero()
        print(f"point {point} wave {wave} run {it}: {len(d)} differ; (b, head, tile):", sorted({(b_, h_, d_ // 16) for b_, h_, d_ in d.tolist()})[:24],
              "max", (o - ref).abs().max().item())
