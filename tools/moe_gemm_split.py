"""W4A16 grouped GEMMs of fused_experts (Mixtral shapes) timed one by one: uniform and routed (ragged) rows per expert."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel  # noqa
dev = "cuda"
E, Hd, I, gs, topk = 8, 4096, 14336, 128, 2


def timeit(f, it=10):
    for _ in range(5): f()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): f()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it


w1 = torch.randint(0, 256, (E, 2 * I, Hd // 2), device=dev, dtype=torch.uint8)
w2 = torch.randint(0, 256, (E, Hd, I // 2), device=dev, dtype=torch.uint8)
s1 = torch.rand(E, 2 * I, Hd // gs, device=dev).to(torch.bfloat16) * 0.01
s2 = torch.rand(E, Hd, I // gs, device=dev).to(torch.bfloat16) * 0.01
import ctypes
# The stamps exist in the DIAGNOSTIC build only (build.py --probes). Run this tool with that build in front of the release
# library so that the Python ops resolve to it:  LD_PRELOAD=sgl-kernel-xpu_amd/build/libsglk_probes.so python tools/moe_gemm_split.py
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so"))
lib.sglk_debug_set_moe_clock_stamps.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(256 * 4 + 256 * 8 * 2, dtype=torch.int32, device=dev)


def clock_of(f):
    lib.sglk_debug_set_moe_clock_stamps(stamps.data_ptr())
    for _ in range(5): f()
    torch.cuda.synchronize()
    lib.sglk_debug_set_moe_clock_stamps(None)
    hst = stamps.cpu()
    st = hst[:1024].view(256, 4).double()
    ok = st[:, 1] > 0
    wv = hst[1024:].view(256, 8, 2).double()[ok] / st[ok][:, 2].view(-1, 1, 1)   # per K block
    st = st[ok]
    return (f"{(100 * st[:, 0] / st[:, 1]).median().item():.0f} MHz, {(st[:, 0] / st[:, 2]).median().item():.0f} cycles per K block; "
            f"per wave and block (diagnostic build of the library only): own-data wait waves 0-3 {wv[:, :4, 0].mean().item():.0f} / 4-7 "
            f"{wv[:, 4:, 0].mean().item():.0f}, barrier wait {wv[:, :4, 1].mean().item():.0f} / {wv[:, 4:, 1].mean().item():.0f}")


if os.environ.get("MOE_WIDE"):  # 0: the 128-row blocks as 128 x 256 tiles (MS = 2) instead of 128 x 512
    lib.sglk_debug_set_moe_persist_wide(int(os.environ["MOE_WIDE"]))
if os.environ.get("MOE_W4_MT"):  # forced tile form of the streaming kernels (12: the K-split kernel, 11: 64-column workgroups, ...)
    lib.sglk_debug_set_w4a16_probe(0, int(os.environ["MOE_W4_MT"]))
if os.environ.get("MOE_PRIO"):
    lib.sglk_debug_set_moe_prio(int(os.environ["MOE_PRIO"]))
for T in (int(a) for a in (sys.argv[1:] or ["2048"])):
    total = T * topk
    ti = torch.randn(T, E, device=dev).topk(topk, dim=-1).indices
    routed = torch.bincount(ti.flatten(), minlength=E).to(torch.int32)
    for name, rows in (("uniform", torch.full((E,), total // E, dtype=torch.int32, device=dev)), ("routed", routed)):
        x = torch.randn(total, Hd, device=dev, dtype=torch.bfloat16) * 0.1
        h = torch.empty(total, I, device=dev, dtype=torch.bfloat16)
        y = torch.empty(total, Hd, device=dev, dtype=torch.bfloat16)
        op = torch.ops.sgl_kernel
        t1 = timeit(lambda: op.moe_grouped_mm_nt_w4a16_act(h, x, w1, s1, None, None, rows, E, True, gs, 1, 0.0))
        t2 = timeit(lambda: op.moe_grouped_mm_nt_xe20_w4a16(y, h, w2, s2, None, None, rows, E, True, gs))
        gu = torch.empty(total, 2 * I, device=dev, dtype=torch.bfloat16)
        t3 = timeit(lambda: op.moe_grouped_mm_nt_xe20_w4a16(gu, x, w1, s1, None, None, rows, E, True, gs))
        del gu
        if total >= 96 * E:
            print("   gate/up:", clock_of(lambda: op.moe_grouped_mm_nt_w4a16_act(h, x, w1, s1, None, None, rows, E, True, gs, 1, 0.0)),
                  "| down:", clock_of(lambda: op.moe_grouped_mm_nt_xe20_w4a16(y, h, w2, s2, None, None, rows, E, True, gs)))
        if os.environ.get("MOE_BF16") and total >= 96 * E:
            wb = (torch.randn(E, 2 * I, Hd, device=dev) * 0.02).to(torch.bfloat16)
            gu = torch.empty(total, 2 * I, device=dev, dtype=torch.bfloat16)
            fb = lambda: op.moe_grouped_mm_nt_xe20(gu, x, wb, None, rows, E, 0, False, 1.702, 7.0)
            tb = timeit(fb)
            print(f"   16-bit weights gate/up (no activation): {tb*1e3:.0f} us;", clock_of(fb)[:60])
            del wb, gu
        fl1, fl2 = 2.0 * total * 2 * I * Hd, 2.0 * total * Hd * I
        print(f"T={T} {name} rows={rows.tolist()}: gate/up {t1*1e3:.0f} us ({fl1/t1/1e9:.0f} TFLOP/s)  down {t2*1e3:.0f} us ({fl2/t2/1e9:.0f} TFLOP/s)  gate/up without the activation {t3*1e3:.0f} us")
