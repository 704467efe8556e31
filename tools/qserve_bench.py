"""qserve_w4a8_per_chn_gemm / per_group_gemm timing (random codes): M sweep at N = K = 4096 and at N = 14336, K = 4096."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
dev = "cuda"


_probes = None
if os.environ.get("QSERVE_CFGS"):  # diagnostic build only (LD_PRELOAD=.../build/libsglk_probes.so): forced stream configurations
    import ctypes
    _probes = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so"))
if os.environ.get("QSERVE_MF"):  # diagnostic build only (LD_PRELOAD=.../build/libsglk_probes.so)
    import ctypes
    ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                ).sglk_debug_set_qserve_mf(int(os.environ["QSERVE_MF"]))


def timeit(f, it=30):
    """device time per call: `it` calls in one HIP graph, median of three replays"""
    for _ in range(10): f()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(it): f()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); g.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) / it)
    return sorted(ts)[1]


shapes = ((4096, 4096), (14336, 4096))
if os.environ.get("QSERVE_SHAPES"):  # e.g. "4096x14336,8192x8192" (N x K)
    shapes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["QSERVE_SHAPES"].split(","))
for N, K in shapes:
    w = torch.randint(-128, 128, (N, K // 2), device=dev, dtype=torch.int8)
    ws = (torch.rand(N, device=dev) * 0.01).half()
    wz = (torch.rand(N, device=dev) * 0.01).half()
    z8 = torch.randint(-8, 8, (K // 128, N), device=dev, dtype=torch.int8)
    s8 = torch.randint(1, 8, (K // 128, N), device=dev, dtype=torch.int8)
    for M in ([int(a) for a in sys.argv[1:]] or [1, 16, 64, 256, 4096]):
        a = torch.randint(-127, 128, (M, K), device=dev, dtype=torch.int8)
        sa = (torch.rand(M, device=dev) * 0.01).half()
        ssum = torch.rand(M, device=dev).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        ms = timeit(lambda: sgl_kernel.qserve_w4a8_per_chn_gemm(a, w, ws, sa, wz, ssum, out))
        ms2 = timeit(lambda: sgl_kernel.qserve_w4a8_per_group_gemm(a, w, z8, s8, ws, sa, out))
        print(f"N={N} K={K} M={M}: per_chn {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TOP/s weights {N*K/2/ms/1e6:.0f} GB/s | "
              f"per_group {ms2*1e3:.1f} us {2.0*M*N*K/ms2/1e9:.1f} TOP/s")
        if _probes is not None and M <= 512:
            ref_c = torch.empty_like(out); ref_g = torch.empty_like(out)
            sgl_kernel.qserve_w4a8_per_chn_gemm(a, w, ws, sa, wz, ssum, ref_c)
            sgl_kernel.qserve_w4a8_per_group_gemm(a, w, z8, s8, ws, sa, ref_g)
            for cfg in [int(c) for c in os.environ["QSERVE_CFGS"].split(",")]:
                _probes.sglk_debug_set_qserve_cfg(cfg)
                o1 = torch.empty_like(out); o2 = torch.empty_like(out)
                sgl_kernel.qserve_w4a8_per_chn_gemm(a, w, ws, sa, wz, ssum, o1)
                sgl_kernel.qserve_w4a8_per_group_gemm(a, w, z8, s8, ws, sa, o2)
                same = bool(torch.equal(o1, ref_c)) and bool(torch.equal(o2, ref_g))
                t1 = timeit(lambda: sgl_kernel.qserve_w4a8_per_chn_gemm(a, w, ws, sa, wz, ssum, out))
                t2 = timeit(lambda: sgl_kernel.qserve_w4a8_per_group_gemm(a, w, z8, s8, ws, sa, out))
                print(f"    cfg {cfg}: per_chn {t1*1e3:.1f} us | per_group {t2*1e3:.1f} us | bit-identical to the default path: {same}")
                _probes.sglk_debug_set_qserve_cfg(0)
