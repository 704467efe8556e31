"""flash_mla_prefill timing: 8-wave kernel against the rows128 kernel (hook value 1 / 0)."""
import ctypes, sys, time, torch
sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("sgl_kernel", "sgl-kernel-xpu_amd/python/sgl_kernel/__init__.py")
sglk = importlib.util.module_from_spec(spec); sys.modules["sgl_kernel"] = sglk; spec.loader.exec_module(sglk)
lib = ctypes.CDLL("sgl-kernel-xpu_amd/python/sgl_kernel/libsglk.so")
dev = "cuda"
for H, bs, sq, sk in ((128, 4, 2048, 2048), (16, 4, 2048, 2048), (128, 16, 512, 4096)):
    page = 64
    nblk = (sk + page - 1) // page
    q_nope = torch.randn(bs * sq, H, 512, device=dev, dtype=torch.bfloat16)
    q_pe = torch.randn(bs * sq, H, 64, device=dev, dtype=torch.bfloat16)
    cache = torch.randn(bs * nblk, page, 576, device=dev, dtype=torch.bfloat16)
    table = torch.arange(bs * nblk, device=dev, dtype=torch.int32).view(bs, nblk)
    cu = torch.arange(bs + 1, device=dev, dtype=torch.int32) * sq
    sl = torch.full((bs,), sk, device=dev, dtype=torch.int32)
    outs = []
    for hook in (1, 0):
        lib.sglk_debug_set_mla_waves_per_group(hook)
        ws = torch.empty(1, device=dev, dtype=torch.uint8)
        f = lambda: sglk.flash_mla_prefill(q_nope, q_pe, cache, cu, sl, sq, table, ws, 0.07, True)
        for _ in range(5): o = f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): o = f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        fl = 2.0 * bs * H * (576 + 512) * (sq * (sk - sq) + sq * (sq + 1) / 2)
        print("H=%d bs=%d sq=%d sk=%d kernel=%s: %.3f ms, %.0f TFLOP/s" % (H, bs, sq, sk, "8-wave" if hook else "rows128", dt * 1e3, fl / dt / 1e12))
        outs.append(o.float())
    print("  max diff between the kernels: %.4g" % (outs[0] - outs[1]).abs().max().item())
lib.sglk_debug_set_mla_waves_per_group(0)
