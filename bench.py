#!/usr/bin/env python3
"""Headline benchmark: per_token_group_quant_fp8 + fp8_blockwise_scaled_mm at the Llama-3-8B FFN
shape (BASELINE.json configs[1]: M=4096, N=14336, K=4096, 128-block scales) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
quantise x[M,K] bf16 -> (e4m3, column-major 1x128 scales), then out = fp8_blockwise_scaled_mm(...)
in bf16. value = whole-job GEMM TFLOP/s (2*M*N*K per step per GPU; the quant pass is inside the timed
region). N > 1: the path does not shard (per-GPU leaf kernels, SURVEY.md section 8e: "replicas only"),
so every rank runs an independent replica; rank 0 prints ONE JSON line. Without a launcher (WORLD_SIZE unset)
`--gpus N` starts the N rank processes itself; under torch.distributed.run it uses the launcher's ranks.

roofline.achieved is measured live with HIP events recorded on the launch stream around each GEMM
launch of the timed region. cpu_baseline times the CPU oracle (torch eager, best of a few thread-pool widths) on a row
sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sgl-kernel-xpu_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

M, N, K = 4096, 14336, 4096
GROUP = 128
FP8 = torch.float8_e4m3fn
PEAK_FP8_TFLOPS = 5000.0  # MI355X_MICROARCH.md: dense FP8 MFMA peak (MX K=128 form), 2:1 sparsity excluded
PEAK_HBM_GBS = 8000.0
GEMM_KERNEL = ("gemm_fp8bw_x32_kernel<bf16> (ONE launch: per workgroup three 256-row tiles, then one 128-row half "
               "tile of the last partial round on a three-stage LDS ring)")
PMC_FILE = os.path.join("profiles", "r05", "bench_pmc.json")
MLA_PMC_FILE = os.path.join("profiles", "r05", "mla_pmc.json")
CLOCK_RAMP_S = 0.15  # untimed steady-state run of the step before the W warm-up steps
REPS = 3             # repetitions of the timed region (each: ramp, W warm-up steps, exactly K timed steps)
BUILD_DIR = os.path.join(ROOT, "sgl-kernel-xpu_amd", "build")


def make_inputs(dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    # weights as tests/test_fp8_blockwise_gemm.py:66-80: uniform over the e4m3 range, [N,K] == column-major [K,N]
    b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev).t()
    sb = (torch.rand(N // 128, K // 128, generator=g) * 1e-3 + 1e-4).to(dev).t()  # column-major [K/128, N/128]
    q = torch.empty(M, K, dtype=FP8, device=dev)
    s = torch.empty(K // GROUP, M, dtype=torch.float32, device=dev).t()  # column-major [M, K/128]
    return x, b, sb, q, s


def pmc_traffic_bytes():
    """HBM-side bytes per GEMM launch from the committed rocprofv3 PMC passes (tools/gpu_profiles_r03.sh ->
    profiles/r03/bench_pmc.json): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts 128-B
    requests as 64 B on gfx950 (MI355X_MICROARCH.md, HBM section) and is doubled."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            pmc = json.load(f)
        # one GEMM = one launch of each instantiation of the persistent kernel (whole tiles, then half tiles)
        ks = [v for name, v in pmc.items() if "gemm_fp8bw_x32_kernel" in name or "persist_kernel" in name]
        return int(sum(2 * k["FETCH_SIZE"]["avg"] + k["WRITE_SIZE"]["avg"] for k in ks) * 1024) if ks else None
    except Exception:
        return None


def mla_pmc_traffic_bytes():
    """The same for flash_mla_decode's kernels (profiles/r04/mla_pmc.json), per op call."""
    try:
        with open(os.path.join(ROOT, MLA_PMC_FILE)) as f:
            pmc = json.load(f)
        ks = [v for name, v in pmc.items() if name.startswith("mla_") or "mla_rows128" in name or "mla_reduce" in name]
        return int(sum(2 * k["FETCH_SIZE"]["avg"] + k["WRITE_SIZE"]["avg"] for k in ks) * 1024) if ks else None
    except Exception:
        return None


def mfma_ceiling(dev):
    """Same-run measured ceiling of the matrix pipes: the register-only stream of tools/mfma_ceiling.hip (the GEMM's own
    MFMA, random e4m3 operands, two waves per SIMD, every CU) behind its own clock ramp; median of 20 launches."""
    import ctypes

    try:
        lib = ctypes.CDLL(os.path.join(BUILD_DIR, "libsglk_ceiling.so"))
    except OSError:
        return None
    lib.sglk_bench_mfma_ceiling.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.sglk_bench_mfma_ceiling_flop.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.sglk_bench_mfma_ceiling_flop.restype = ctypes.c_double
    blocks = torch.cuda.get_device_properties(dev).multi_processor_count
    iters = 1500  # ~0.2 ms per launch: the length of the GEMM
    g = torch.Generator(device="cpu").manual_seed(7)
    src = torch.randint(0, 256, (16384 * 4,), generator=g, dtype=torch.uint8)
    src[(src & 0x7F) == 0x7F] ^= 1  # no e4m3 NaN codes
    src = src.to(dev)
    dst = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.sglk_bench_mfma_ceiling(stream, src.data_ptr(), dst.data_ptr(), blocks, iters)
        if rc:
            raise RuntimeError("mfma ceiling launch failed: %d" % rc)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < CLOCK_RAMP_S:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b_ in ev:
        a.record()
        run()
        b_.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b_) for a, b_ in ev)
    tflops = lib.sglk_bench_mfma_ceiling_flop(blocks, iters) / (ms[len(ms) // 2] * 1e-3) / 1e12
    return {"tflops": round(tflops, 1),
            "kernel": "register-only v_mfma_scale_f32_32x32x64_f8f6f4 stream, random e4m3 operands, 2 waves per SIMD, "
                      "%d workgroups (tools/mfma_ceiling.hip), same process and device" % blocks,
            "frac_of_nominal": round(tflops / PEAK_FP8_TFLOPS, 4)}


def mfma_ceiling_bf16(dev, waves):
    """The same for the bf16 kernels: a register-only v_mfma_f32_32x32x16_bf16 stream on random operands, `waves` waves per
    workgroup (4 = one per SIMD as flash_mla_decode's 128-row kernel, 8 = two as the prefill / MoE kernels), one workgroup
    per CU, behind its own clock ramp; median of 20 launches. TFLOP/s, or None without the library."""
    import ctypes

    try:
        lib = ctypes.CDLL(os.path.join(BUILD_DIR, "libsglk_ceiling.so"))
        fn, flop = lib.sglk_bench_mfma_ceiling_bf16, lib.sglk_bench_mfma_ceiling_bf16_flop
    except (OSError, AttributeError):
        return None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    flop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    flop.restype = ctypes.c_double
    blocks = torch.cuda.get_device_properties(dev).multi_processor_count
    iters = 3000 if waves == 4 else 1500  # ~0.2 - 0.3 ms per launch
    g = torch.Generator(device="cpu").manual_seed(9)
    src = torch.randint(0, 2 ** 31 - 1, (16384,), generator=g, dtype=torch.int32).to(dev)
    dst = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        rc = fn(stream, src.data_ptr(), dst.data_ptr(), blocks, waves, iters)
        if rc:
            raise RuntimeError("bf16 mfma ceiling launch failed: %d" % rc)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < CLOCK_RAMP_S:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b_ in ev:
        a.record()
        run()
        b_.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b_) for a, b_ in ev)
    return round(flop(blocks, waves, iters) / (ms[len(ms) // 2] * 1e-3) / 1e12, 1)


def gemm_clock(q, b, s, sb, dev):
    """Median shader clock (MHz) and shader cycles per K block of the whole-tile phase of the GEMM, from the stamps the
    workgroups of the DIAGNOSTIC build of the library write (build/libsglk_probes.so, build.py --probes; the release
    library has no such code). Called through the C-ABI with the tensors' pointers. None if that build is absent."""
    import ctypes

    try:
        lib = ctypes.CDLL(os.path.join(BUILD_DIR, "libsglk_probes.so"))
        arm = lib.sglk_debug_set_gemm_stamps
        mm = lib.sglk_fp8_blockwise_scaled_mm
    except (OSError, AttributeError):
        return None
    arm.argtypes = [ctypes.c_void_p]
    arm.restype = None
    mm.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] * 10 + [ctypes.c_int]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        rc = mm(stream, out.data_ptr(), q.data_ptr(), b.data_ptr(), s.data_ptr(), sb.data_ptr(), M, N, K, q.stride(0),
                b.stride(1), out.stride(0), s.stride(0), s.stride(1), sb.stride(0), sb.stride(1), 2)  # 2 = SGLK_BF16 (include/sglk.h)
        if rc:
            raise RuntimeError("probes build: fp8_blockwise_scaled_mm failed (%d)" % rc)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < CLOCK_RAMP_S:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    # (a workgroup writes its whole-tile record at index blockIdx.x and its half-tile record at blockIdx.x + 256: the grid is
    #  one workgroup per CU, at most 256 on this part - sized for that, and checked)
    assert torch.cuda.get_device_properties(dev).multi_processor_count <= 256, "stamp buffer is sized for <= 256 workgroups"
    buf = torch.zeros(512 * 4, dtype=torch.int32, device=dev)
    arm(buf.data_ptr())
    try:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    finally:
        arm(None)
    st = buf.cpu().view(512, 4).to(torch.float64)
    whole, half = st[(st[:, 1] > 0) & (st[:, 3] == 4)], st[(st[:, 1] > 0) & (st[:, 3] == 2)]
    if whole.numel() == 0:
        return None
    res = {"mhz": round((100.0 * whole[:, 0] / whole[:, 1]).median().item()),
           "cycles_per_k_block": round((whole[:, 0] / whole[:, 2]).median().item())}
    if half.numel():
        res["half_tile_mhz"] = round((100.0 * half[:, 0] / half[:, 1]).median().item())
        res["half_tile_cycles_per_k_block"] = round((half[:, 0] / half[:, 2]).median().item())
    return res


def cpu_baseline(seconds_budget=20.0):
    """CPU oracle on a bounded row sample of the same workload (same N, K; fewer rows)."""
    from oracle import gemm as ogemm
    from oracle import quant as oquant

    rows = 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(rows, K, generator=g).to(torch.bfloat16)
    b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).t()
    sb = (torch.rand(N // 128, K // 128, generator=g) * 1e-3 + 1e-4).t()
    # The GPU boxes are shared 256-core hosts: a pool as wide as the host is not the fastest one for this oracle. Time a few
    # widths within the budget and report the best, with the width it ran at.
    ncpu = os.cpu_count() or 1
    widths = sorted({min(w, ncpu) for w in (16, 64, ncpu)})
    best, before = None, torch.get_num_threads()
    for w in widths:
        torch.set_num_threads(w)
        t0 = time.perf_counter()
        reps = 0
        while True:
            q, s, _ = oquant.per_token_group_quant_8bit(x, GROUP, FP8)
            ogemm.fp8_blockwise_scaled_mm(q, b, s, sb, torch.bfloat16)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > seconds_budget / len(widths) or reps >= 3:
                break
        tflops = 2.0 * rows * N * K * reps / dt / 1e12
        if best is None or tflops > best[0]:
            best = (tflops, w, reps)
    torch.set_num_threads(before)
    return {
        "value": round(best[0], 4),
        "unit": "TFLOP/s",
        "cores": best[1],       # threads the reported pass actually used (the contract's meaning of the field)
        "threads": best[1],
        "host_cores": ncpu,     # what the host has (shared with other tenants on this pool)
        "kind": "port",
        "sample": f"{best[2]} pass(es) of quant+GEMM on {rows} of {M} rows (same N={N}, K={K}), torch-eager oracle, "
                  f"best of thread-pool widths {widths}",
    }


def side_metrics(sgl_kernel, dev):
    """HBM-bound companions of the path (BASELINE configs[0] shapes), reported as extra evidence."""
    out = {}
    eager_legs = []  # (timing mode is kept apart from the metrics: legs whose graph capture failed are listed at the end)

    def timeit(fn, iters=30):
        """Device time per call: `iters` calls captured into ONE HIP graph and one replay of it timed with events (the
        decode-sized kernels here take 5 - 20 us, less than a Python op call costs on a busy host: an eager loop would time
        the host). Every op of the library is capture-safe by contract; if a capture fails the eager loop is timed."""
        for _ in range(max(5, iters)):  # the chip's clocks ramp for tens of milliseconds
            fn()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(iters):
                        fn()
            torch.cuda.current_stream().wait_stream(side)
            graph.replay()
            torch.cuda.synchronize()
            st.record()
            graph.replay()
            en.record()
            torch.cuda.synchronize()
            del graph
            return st.elapsed_time(en) / iters
        except Exception:
            torch.cuda.synchronize()
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        torch.cuda.synchronize()
        eager_legs.append(len(out))
        return st.elapsed_time(en) / iters

    x = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    w = torch.randn(4096, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    ms = timeit(lambda: sgl_kernel.rmsnorm(x, w, 1e-6, out=y))
    out["rmsnorm_4096x4096_bf16_GBs"] = round((2 * x.numel() * 2 + 4096 * 2) / ms / 1e6, 1)
    x2 = torch.randn(4096, 8192, device=dev, dtype=torch.bfloat16)
    o2 = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: sgl_kernel.silu_and_mul(x2, out=o2))
    out["silu_and_mul_4096x8192_bf16_GBs"] = round(3 * o2.numel() * 2 / ms / 1e6, 1)
    # the two swiglu variants of fused_experts on the same stream of bytes (two reads + one write per output element)
    ms = timeit(lambda: sgl_kernel.silu_and_mul_clamp(x2, o2, 10.0))
    out["silu_and_mul_clamp_4096x8192_bf16_GBs"] = round(3 * o2.numel() * 2 / ms / 1e6, 1)
    ms = timeit(lambda: sgl_kernel.swiglu_gpt_oss_sigmoid_alpha(x2, 1.702, 7.0))
    out["swiglu_gpt_oss_4096x8192_bf16_GBs"] = round(3 * o2.numel() * 2 / ms / 1e6, 1)
    q = torch.empty(4096, 4096, dtype=FP8, device=dev)
    s = torch.empty(32, 4096, dtype=torch.float32, device=dev).t()
    ms = timeit(lambda: sgl_kernel.sgl_per_token_group_quant_8bit(x, q, s, 128, 1e-10, -448.0, 448.0, False,
                                                                   enable_v2=False))
    out["per_token_group_quant_fp8_4096x4096_GBs"] = round((x.numel() * 3 + s.numel() * 4) / ms / 1e6, 1)
    # per-token / per-tensor fp8 quantisation of the same activations
    st = torch.zeros(4096, dtype=torch.float32, device=dev)
    ms = timeit(lambda: sgl_kernel.sgl_per_token_quant_fp8(x, q, st))
    out["per_token_quant_fp8_4096x4096_GBs"] = round((x.numel() * 3 + 4096 * 4) / ms / 1e6, 1)
    s1 = torch.zeros(1, dtype=torch.float32, device=dev)
    ms = timeit(lambda: sgl_kernel.sgl_per_tensor_quant_fp8(x, q, s1, False))
    out["per_tensor_quant_fp8_dynamic_4096x4096_GBs"] = round(x.numel() * 5 / ms / 1e6, 1)  # two reads + one write
    del x, x2, o2, q, s, y
    # the M sweep of BASELINE configs[1] (decode .. prefill rows against the same N=14336, K=4096 weights)
    g = torch.Generator(device="cpu").manual_seed(3)
    bw = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev).t()
    sbw = (torch.rand(N // 128, K // 128, generator=g) * 1e-3 + 1e-4).to(dev).t()
    for m in (1, 16, 64, 128, 256, 1024):
        am = ((torch.rand(m, K, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev)
        sam = (torch.rand(K // 128, m, generator=g) * 1e-3 + 1e-4).to(dev).t()
        ms = timeit(lambda: sgl_kernel.fp8_blockwise_scaled_mm(am, bw, sam, sbw, torch.bfloat16), iters=50)
        out[f"fp8_blockwise_gemm_M{m}_us"] = round(ms * 1e3, 1)
        out[f"fp8_blockwise_gemm_M{m}_weight_GBs"] = round(N * K / ms / 1e6, 1)
        if m <= 64:  # fp8_scaled_mm (per-row / per-column scales) on the same weights
            sa1 = torch.rand(m, 1, device=dev) * 1e-3 + 1e-4
            sb1 = torch.rand(N, 1, device=dev) * 1e-3 + 1e-4
            ms = timeit(lambda: sgl_kernel.fp8_scaled_mm(am, bw, sa1, sb1, torch.bfloat16), iters=50)
            out[f"fp8_scaled_mm_M{m}_us"] = round(ms * 1e3, 1)
    # the down projection of the same layer (N = 4096, K = 14336) at decode-batch .. chunk rows: K-slice units since round 5
    # (50 - 96 us unsliced for the block-scale mode, 41 - 117 for the row / column scale modes)
    bwd = ((torch.rand(K, N, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev).t()  # [N -> K rows of 14336] as [14336, 4096]^T
    sbd = (torch.rand(K // 128, N // 128, generator=g) * 1e-3 + 1e-4).to(dev).t()
    bid = torch.randint(-127, 128, (K, N), generator=g, dtype=torch.int8).to(dev).t()
    sbcol = torch.rand(K, 1, device=dev) * 1e-3 + 1e-4
    for m in (128, 256, 512, 1024):
        amd = ((torch.rand(m, N, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev)
        samd = (torch.rand(N // 128, m, generator=g) * 1e-3 + 1e-4).to(dev).t()
        ms = timeit(lambda: sgl_kernel.fp8_blockwise_scaled_mm(amd, bwd, samd, sbd, torch.bfloat16), iters=30)
        out[f"fp8_blockwise_gemm_down_N4096_K14336_M{m}_us"] = round(ms * 1e3, 1)
        sarow = torch.rand(m, 1, device=dev) * 1e-3 + 1e-4
        ms = timeit(lambda: sgl_kernel.fp8_scaled_mm(amd, bwd, sarow, sbcol, torch.bfloat16), iters=30)
        out[f"fp8_scaled_mm_down_N4096_K14336_M{m}_us"] = round(ms * 1e3, 1)
        aid = torch.randint(-127, 128, (m, N), generator=g, dtype=torch.int8).to(dev)
        ms = timeit(lambda: sgl_kernel.int8_scaled_mm(aid, bid, sarow, sbcol, torch.bfloat16), iters=30)
        out[f"int8_scaled_mm_down_N4096_K14336_M{m}_us"] = round(ms * 1e3, 1)
    del bwd, bid
    # fp8_scaled_mm / int8_scaled_mm at the headline shape (row x column scales: the persistent pipeline without block scales)
    am = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev)
    sa1 = torch.rand(M, 1, device=dev) * 1e-3 + 1e-4
    sb1 = torch.rand(N, 1, device=dev) * 1e-3 + 1e-4
    ms = timeit(lambda: sgl_kernel.fp8_scaled_mm(am, bw, sa1, sb1, torch.bfloat16), iters=30)
    out["fp8_scaled_mm_M4096_TFLOPs"] = round(2.0 * M * N * K / ms / 1e9, 1)
    ai = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
    bi = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev).t()
    ms = timeit(lambda: sgl_kernel.int8_scaled_mm(ai, bi, sa1, sb1, torch.bfloat16), iters=30)
    out["int8_scaled_mm_M4096_TOPs"] = round(2.0 * M * N * K / ms / 1e9, 1)
    # QServe W4A8 at the same shape (random codes in the QServe packing; group scales 1..7, zero terms -8..7)
    wq = torch.randint(-128, 128, (N, K // 2), generator=g, dtype=torch.int8).to(dev)
    ws16 = (torch.rand(N, device=dev) * 0.01).half()
    wz16 = (torch.rand(N, device=dev) * 0.01).half()
    z8 = torch.randint(-8, 8, (K // 128, N), generator=g, dtype=torch.int8).to(dev)
    s8 = torch.randint(1, 8, (K // 128, N), generator=g, dtype=torch.int8).to(dev)
    sa16 = (torch.rand(M, device=dev) * 0.01).half()
    ssum = torch.rand(M, device=dev).half()
    qo = torch.empty(M, N, device=dev, dtype=torch.float16)
    ms = timeit(lambda: sgl_kernel.qserve_w4a8_per_chn_gemm(ai, wq, ws16, sa16, wz16, ssum, qo), iters=30)
    out["qserve_w4a8_per_chn_M4096_TOPs"] = round(2.0 * M * N * K / ms / 1e9, 1)
    ms = timeit(lambda: sgl_kernel.qserve_w4a8_per_group_gemm(ai, wq, z8, s8, ws16, sa16, qo), iters=30)
    out["qserve_w4a8_per_group_M4096_TOPs"] = round(2.0 * M * N * K / ms / 1e9, 1)
    for m in (1, 64, 128):  # (128 rows: on the 32x32x32 stream kernel since round 5 - 56 us on the 128 x 128 tile kernel before)
        ms = timeit(lambda: sgl_kernel.qserve_w4a8_per_group_gemm(ai[:m], wq, z8, s8, ws16, sa16[:m], qo[:m]), iters=50)
        out[f"qserve_w4a8_per_group_M{m}_us"] = round(ms * 1e3, 1)
        out[f"qserve_w4a8_per_group_M{m}_weight_GBs"] = round(N * K / 2 / ms / 1e6, 1)
    del bw, sbw, ai, bi, am, wq, qo
    # the down projection of the same layer at decode: N=4096, K=14336 (few n-tiles: K split over the waves of a workgroup)
    bw = ((torch.rand(K, N, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev).t()
    sbw = (torch.rand(K // 128, N // 128, generator=g) * 1e-3 + 1e-4).to(dev).t()
    for m in (1, 16, 64):
        am = ((torch.rand(m, N, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(FP8).to(dev)
        sam = (torch.rand(N // 128, m, generator=g) * 1e-3 + 1e-4).to(dev).t()
        ms = timeit(lambda: sgl_kernel.fp8_blockwise_scaled_mm(am, bw, sam, sbw, torch.bfloat16), iters=50)
        out[f"fp8_blockwise_gemm_down_M{m}_us"] = round(ms * 1e3, 1)
    del bw, sbw
    # flash_mla_decode, BASELINE configs[3]: bs=128, seq=8192, kv_lora 512 + rope 64, paged (64), bf16.
    # Bytes as benchmark/bench_flash_mla_decode.py:109-115 of the reference: q + kv cache + table + seq_lens + out.
    bs, seq, page = 128, 8192, 64
    n_pages = seq // page
    cache = torch.randn(bs * n_pages, page, 576, device=dev, dtype=torch.bfloat16)
    table = torch.randint(0, bs * n_pages, (bs, n_pages), device=dev, dtype=torch.int32)
    seq_lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
    for H in (128, 64, 32, 16):  # the head counts of the reference benchmark (bench_flash_mla_decode.py:26-35)
        qq = torch.randn(bs, H, 576, device=dev, dtype=torch.bfloat16) * 100
        q_nope, q_pe = qq[..., :512], qq[..., 512:].contiguous()
        ws = torch.empty(sgl_kernel.flash_mla_get_workspace_size(seq, bs, H, page, -1), device=dev, dtype=torch.uint8)
        ms = timeit(lambda: sgl_kernel.flash_mla_decode(q_nope, q_pe, cache, seq_lens, table, ws, 576 ** -0.5, -1),
                    iters=50)
        nbytes = qq.numel() * 2 + cache.numel() * 2 + table.numel() * 4 + seq_lens.numel() * 4 + bs * H * 512 * 2
        out[f"flash_mla_decode_bs128_seq8192_h{H}_GBs"] = round(nbytes / ms / 1e6, 1)
        out[f"flash_mla_decode_bs128_seq8192_h{H}_ms"] = round(ms, 4)
        out[f"flash_mla_decode_bs128_seq8192_h{H}_TFLOPs"] = round(2.0 * bs * H * seq * (576 + 512) / ms / 1e9, 1)
    # the same with 128-token pages (bench_flash_mla_decode.py sweeps page sizes 64 and 128)
    cache128 = cache.view(bs * n_pages // 2, 128, 576)
    table128 = torch.randint(0, bs * n_pages // 2, (bs, n_pages // 2), device=dev, dtype=torch.int32)
    qq = torch.randn(bs, 128, 576, device=dev, dtype=torch.bfloat16) * 100
    q_nope, q_pe = qq[..., :512], qq[..., 512:].contiguous()
    ws = torch.empty(sgl_kernel.flash_mla_get_workspace_size(seq, bs, 128, 128, -1), device=dev, dtype=torch.uint8)
    ms = timeit(lambda: sgl_kernel.flash_mla_decode(q_nope, q_pe, cache128, seq_lens, table128, ws, 576 ** -0.5, -1), iters=50)
    nbytes = qq.numel() * 2 + cache.numel() * 2 + table128.numel() * 4 + seq_lens.numel() * 4 + bs * 128 * 512 * 2
    out["flash_mla_decode_bs128_seq8192_h128_page128_GBs"] = round(nbytes / ms / 1e6, 1)
    out["flash_mla_decode_bs128_seq8192_h128_page128_ms"] = round(ms, 4)
    del cache, table, seq_lens, cache128, table128
    # flash_mla_prefill (same latent cache layout): 16 sequences, 512 new tokens each over 4096 cached keys, causal
    pb, psq, psk = 16, 512, 4096
    pcache = torch.randn(pb * psk // page, page, 576, device=dev, dtype=torch.bfloat16)
    ptable = torch.arange(pb * psk // page, device=dev, dtype=torch.int32).view(pb, psk // page)
    pq = torch.randn(pb * psq, 128, 576, device=dev, dtype=torch.bfloat16)
    pqn, pqp = pq[..., :512], pq[..., 512:].contiguous()
    pcu = torch.arange(pb + 1, device=dev, dtype=torch.int32) * psq
    psl = torch.full((pb,), psk, device=dev, dtype=torch.int32)
    pws = torch.empty(1, device=dev, dtype=torch.uint8)
    ms = timeit(lambda: sgl_kernel.flash_mla_prefill(pqn, pqp, pcache, pcu, psl, psq, ptable, pws, 576 ** -0.5, True),
                iters=5)
    out["flash_mla_prefill_h128_16x512_over_4096_ms"] = round(ms, 4)
    out["flash_mla_prefill_h128_16x512_over_4096_TFLOPs"] = round(
        2.0 * pb * 128 * (576 + 512) * (psq * (psk - psq) + psq * (psq + 1) / 2) / ms / 1e9, 1)
    del pcache, ptable, pq, pqn, pqp
    # fwd (flash attention), BASELINE configs[2]: bs=16, 32 q heads / 8 kv heads, d=128, seq=4096, paged (64), bf16.
    from sgl_kernel.flash_attn import flash_attn_with_kvcache

    bs, hq, hk, d, seq, page = 16, 32, 8, 128, 4096, 64
    n_pages = bs * seq // page
    kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
    lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
    qd = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, page_table=pt, causal=True), iters=20)
    out["fwd_decode_bs16_h32_kv8_d128_seq4096_GBs"] = round((kc.numel() + vc.numel() + 2 * qd.numel()) * 2 / ms / 1e6, 1)
    out["fwd_decode_bs16_h32_kv8_d128_seq4096_ms"] = round(ms, 4)
    qp = torch.randn(bs * seq, hq, d, device=dev, dtype=torch.bfloat16)
    cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * seq
    ms = timeit(lambda: flash_attn_with_kvcache(qp, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                max_seqlen_q=seq, causal=True), iters=5)
    out["fwd_prefill_causal_bs16_h32_kv8_d128_seq4096_TFLOPs"] = round(4.0 * bs * hq * d * seq * seq / 2 / ms / 1e9, 1)
    out["fwd_prefill_causal_bs16_h32_kv8_d128_seq4096_ms"] = round(ms, 4)
    # chunked prefill: 128 new tokens per sequence over the 4096 cached keys (bench_flash_attn.py's q = 128 leg)
    qc = torch.randn(bs * 128, hq, d, device=dev, dtype=torch.bfloat16)
    cuc = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * 128
    ms = timeit(lambda: flash_attn_with_kvcache(qc, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cuc,
                                                max_seqlen_q=128, causal=True), iters=20)
    out["fwd_chunk_prefill_q128_bs16_h32_kv8_d128_seq4096_ms"] = round(ms, 4)
    out["fwd_chunk_prefill_q128_bs16_h32_kv8_d128_seq4096_TFLOPs"] = round(
        4.0 * bs * hq * d * (128 * (seq - 128) + 128 * 129 / 2) / ms / 1e9, 1)
    # decode with 128-token pages (the reference benchmark sweeps page sizes 64 and 128)
    kc128, vc128 = kc.view(n_pages // 2, 128, hk, d), vc.view(n_pages // 2, 128, hk, d)
    pt128 = torch.randperm(n_pages // 2, device=dev).to(torch.int32).view(bs, seq // 128)
    ms = timeit(lambda: flash_attn_with_kvcache(qd, kc128, vc128, cache_seqlens=lens, page_table=pt128, causal=True), iters=20)
    out["fwd_decode_bs16_h32_kv8_d128_seq4096_page128_GBs"] = round((kc.numel() + vc.numel() + 2 * qd.numel()) * 2 / ms / 1e6, 1)
    del kc, vc, qp, qd, qc, kc128, vc128
    # prefill at head dim 64 (the reference's other prefill instantiation): same batch / heads / length
    kc64 = torch.randn(n_pages, page, hk, 64, device=dev, dtype=torch.bfloat16)
    vc64 = torch.randn(n_pages, page, hk, 64, device=dev, dtype=torch.bfloat16)
    qp64 = torch.randn(bs * seq, hq, 64, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: flash_attn_with_kvcache(qp64, kc64, vc64, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                max_seqlen_q=seq, causal=True), iters=5)
    out["fwd_prefill_causal_bs16_h32_kv8_d64_seq4096_TFLOPs"] = round(4.0 * bs * hq * 64 * seq * seq / 2 / ms / 1e9, 1)
    out["fwd_prefill_causal_bs16_h32_kv8_d64_seq4096_ms"] = round(ms, 4)
    del kc64, vc64, qp64
    # the same d = 128 prefill with a softcap (Gemma-2) and over an fp8 e4m3 cache (reference tests/test_flash_attention.py:
    # 1691-1704): both on the 128-row-block kernel since round 5
    kcb = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    vcb = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    qpb = torch.randn(bs * seq, hq, d, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: flash_attn_with_kvcache(qpb, kcb, vcb, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                max_seqlen_q=seq, causal=True, softcap=50.0), iters=5)
    out["fwd_prefill_causal_softcap_bs16_h32_kv8_d128_seq4096_TFLOPs"] = round(4.0 * bs * hq * d * seq * seq / 2 / ms / 1e9, 1)
    kc8, vc8, one = kcb.to(FP8), vcb.to(FP8), torch.ones(1, device=dev)
    del kcb, vcb
    ms = timeit(lambda: flash_attn_with_kvcache(qpb, kc8, vc8, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                max_seqlen_q=seq, causal=True, k_descale=one, v_descale=one), iters=5)
    out["fwd_prefill_causal_fp8kv_bs16_h32_kv8_d128_seq4096_TFLOPs"] = round(4.0 * bs * hq * d * seq * seq / 2 / ms / 1e9, 1)
    del kc8, vc8, qpb
    # prefill at head dim 256 (Gemma; reference instantiation FMHAPrefillXe20.cmake:30-54), 16 q heads / 8 kv heads: on the
    # 128-row-block kernel since round 5 (before: the general 16-row kernel, 192 TFLOP/s)
    kc256 = torch.randn(n_pages, page, hk, 256, device=dev, dtype=torch.bfloat16)
    vc256 = torch.randn(n_pages, page, hk, 256, device=dev, dtype=torch.bfloat16)
    qp256 = torch.randn(bs * seq, 16, 256, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: flash_attn_with_kvcache(qp256, kc256, vc256, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                max_seqlen_q=seq, causal=True), iters=5)
    out["fwd_prefill_causal_bs16_h16_kv8_d256_seq4096_TFLOPs"] = round(4.0 * bs * 16 * 256 * seq * seq / 2 / ms / 1e9, 1)
    out["fwd_prefill_causal_bs16_h16_kv8_d256_seq4096_ms"] = round(ms, 4)
    del kc256, vc256, qp256
    # head dims 96 / 192 (reference instantiations FMHAPrefillXe20.cmake:30-54) inside the 128 / 256 LDS images of the same kernel
    # since round 5 (before: the general 16-row kernel)
    for dd, hqd in ((96, 32), (192, 16)):
        kcd = torch.randn(n_pages, page, hk, dd, device=dev, dtype=torch.bfloat16)
        vcd = torch.randn(n_pages, page, hk, dd, device=dev, dtype=torch.bfloat16)
        qpd = torch.randn(bs * seq, hqd, dd, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: flash_attn_with_kvcache(qpd, kcd, vcd, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                    max_seqlen_q=seq, causal=True), iters=5)
        out[f"fwd_prefill_causal_bs16_h{hqd}_kv8_d{dd}_seq4096_TFLOPs"] = round(4.0 * bs * hqd * dd * seq * seq / 2 / ms / 1e9, 1)
        del kcd, vcd, qpd
    # decode at the other head dims / an fp8 KV cache the reference instantiates (FMHADecodeXe20.cmake:13-16, :62-111)
    for dd, kvdt in ((64, torch.bfloat16), (256, torch.bfloat16), (128, FP8)):
        npg = bs * seq // page
        kcd = torch.randn(npg, page, hk, dd, device=dev, dtype=torch.bfloat16).to(kvdt)
        vcd = torch.randn(npg, page, hk, dd, device=dev, dtype=torch.bfloat16).to(kvdt)
        ptd = torch.randperm(npg, device=dev).to(torch.int32).view(bs, seq // page)
        qdd = torch.randn(bs, 1, hq, dd, device=dev, dtype=torch.bfloat16)
        kw = {}
        if kvdt == FP8:
            kw = dict(k_descale=torch.ones(1, device=dev), v_descale=torch.ones(1, device=dev))
        ms = timeit(lambda: flash_attn_with_kvcache(qdd, kcd, vcd, cache_seqlens=lens, page_table=ptd, causal=True, **kw), iters=20)
        tag = f"d{dd}" + ("_fp8kv" if kvdt == FP8 else "")
        out[f"fwd_decode_bs16_h32_kv8_{tag}_seq4096_GBs"] = round(
            ((kcd.numel() + vcd.numel()) * kcd.element_size() + 2 * qdd.numel() * 2) / ms / 1e6, 1)
        out[f"fwd_decode_bs16_h32_kv8_{tag}_seq4096_ms"] = round(ms, 4)
        del kcd, vcd
    # a chunk of ONE long sequence (128 queries over 32768 keys): KV splits in the 128-row-block kernel since round 5 (654 us unsplit)
    ctx = 32768
    kcl = torch.randn(ctx // page, page, hk, d, device=dev, dtype=torch.bfloat16)
    vcl = torch.randn(ctx // page, page, hk, d, device=dev, dtype=torch.bfloat16)
    ptl = torch.randperm(ctx // page, device=dev).to(torch.int32).view(1, ctx // page)
    ql = torch.randn(128, hq, d, device=dev, dtype=torch.bfloat16)
    lens1 = torch.full((1,), ctx, device=dev, dtype=torch.int32)
    cu1 = torch.tensor([0, 128], device=dev, dtype=torch.int32)
    ms = timeit(lambda: flash_attn_with_kvcache(ql, kcl, vcl, cache_seqlens=lens1, page_table=ptl, cu_seqlens_q=cu1,
                                                max_seqlen_q=128, causal=True), iters=10)
    out["fwd_chunk_prefill_bs1_q128_ctx32768_us"] = round(ms * 1e3, 1)
    del kcl, vcl
    # sampling at a Llama-3 vocabulary (reference benchmark: none; tests/test_sampling.py sizes), decode batch sizes
    for sb_ in (1, 64):
        probs = torch.softmax(torch.randn(sb_, 128256, device=dev), dim=-1)
        kk = torch.full((sb_,), 50, device=dev, dtype=torch.int32)
        pp = torch.full((sb_,), 0.9, device=dev, dtype=torch.float32)
        ms = timeit(lambda: sgl_kernel.top_k_renorm_prob(probs, kk), iters=10)
        out[f"top_k_renorm_probs_bs{sb_}_vocab128256_us"] = round(ms * 1e3, 1)
        ms = timeit(lambda: sgl_kernel.top_k_top_p_sampling_from_probs(probs, kk, pp), iters=10)
        out[f"top_k_top_p_sampling_bs{sb_}_vocab128256_us"] = round(ms * 1e3, 1)
        ms = timeit(lambda: sgl_kernel.top_k_top_p_sampling_from_probs(probs, kk, pp, filter_apply_order="joint"), iters=10)
        out[f"top_k_top_p_sampling_joint_bs{sb_}_vocab128256_us"] = round(ms * 1e3, 1)
        ms = timeit(lambda: sgl_kernel.min_p_sampling_from_probs(probs, pp * 0.1), iters=10)
        out[f"min_p_sampling_bs{sb_}_vocab128256_us"] = round(ms * 1e3, 1)
    # qk-norm + rope over a 4096-token chunk (32 + 8 heads of 128): cached angles / computed angles
    from sgl_kernel import elementwise as _ew
    q3 = torch.randn(4096, hq, d, device=dev, dtype=torch.bfloat16)
    k3 = torch.randn(4096, hk, d, device=dev, dtype=torch.bfloat16)
    wn = torch.ones(d, device=dev, dtype=torch.bfloat16)
    cs32 = torch.randn(8192, d, device=dev, dtype=torch.float32)
    posi = torch.randint(0, 8192, (4096,), device=dev)
    ms = timeit(lambda: _ew.fused_inplace_qknorm_rope(q3, k3, wn, wn, cs32, posi, True), iters=30)
    out["fused_inplace_qknorm_rope_T4096_h32_kv8_d128_GBs"] = round(2.0 * (q3.numel() + k3.numel()) * 2 / ms / 1e6, 1)
    qkv = torch.randn(4096, (hq + 2 * hk) * d, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: _ew.fused_qk_norm_rope(qkv, hq, hk, hk, d, 1e-6, wn, wn, 10000.0, True, posi.int()), iters=30)
    out["fused_qk_norm_rope_T4096_h32_kv8_d128_GBs"] = round(2.0 * (q3.numel() + k3.numel()) * 2 / ms / 1e6, 1)
    del q3, k3, qkv
    # MoE routing latencies (reference benchmark/bench_moe_align_block_size.py, bench_moe_topk_softmax.py)
    for toks, ne, tk in ((4096, 8, 2), (4096, 256, 8), (16384, 256, 8)):
        logits = torch.randn(toks, ne, device=dev, dtype=torch.float32)
        tw = torch.empty(toks, tk, device=dev, dtype=torch.float32)
        ti = torch.empty(toks, tk, device=dev, dtype=torch.int32)
        ms = timeit(lambda: sgl_kernel.topk_softmax(tw, ti, logits, True), iters=50)
        out[f"topk_softmax_T{toks}_E{ne}_top{tk}_us"] = round(ms * 1e3, 1)
        for blk in (64, 128):
            max_pad = toks * tk + ne * (blk - 1)
            sorted_ids = torch.empty(max_pad, device=dev, dtype=torch.int32)
            expert_ids = torch.empty((max_pad + blk - 1) // blk, device=dev, dtype=torch.int32)
            n_post = torch.empty(1, device=dev, dtype=torch.int32)
            cumsum = torch.zeros(ne + 1, device=dev, dtype=torch.int32)
            ms = timeit(lambda: sgl_kernel.moe_align_block_size(ti, ne, blk, sorted_ids, expert_ids, n_post, cumsum), iters=50)
            out[f"moe_align_block_size_T{toks}_E{ne}_top{tk}_block{blk}_us"] = round(ms * 1e3, 1)
    # fused_experts int4 W4A16, BASELINE configs[4]: Mixtral-8x7B (8 experts, top-2, hidden 4096, inter 14336, group 128)
    E, Hd, I, gs, topk = 8, 4096, 14336, 128, 2
    w1 = torch.randint(0, 256, (E, 2 * I, Hd // 2), device=dev, dtype=torch.uint8)
    w2 = torch.randint(0, 256, (E, Hd, I // 2), device=dev, dtype=torch.uint8)
    s1 = torch.rand(E, 2 * I, Hd // gs, device=dev).to(torch.bfloat16) * 0.01
    s2 = torch.rand(E, Hd, I // gs, device=dev).to(torch.bfloat16) * 0.01
    for T in (1, 32, 64, 512, 2048, 4096):  # (the reference benchmark sweeps 1 .. 4096 tokens, bench_fused_experts_w4a16.py:459)
        xx = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
        logits = torch.randn(T, E, device=dev, dtype=torch.bfloat16)
        tw = torch.empty(T, topk, device=dev, dtype=torch.float32)
        ti = torch.empty(T, topk, device=dev, dtype=torch.int32)
        sgl_kernel.topk_softmax(tw, ti, logits, True)
        ms = timeit(lambda: sgl_kernel.fused_experts(xx, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2),
                    iters=40 if T <= 256 else 8)
        out[f"fused_experts_w4a16_mixtral_T{T}_ms"] = round(ms, 4)
        out[f"fused_experts_w4a16_mixtral_T{T}_TFLOPs"] = round(2.0 * T * topk * 3 * Hd * I / ms / 1e9, 1)
        # (weights of the experts the tokens actually reach - SURVEY 8(d): min(E, distinct experts) - T = 1 touches two)
        hit = int(torch.unique(ti).numel())
        out[f"fused_experts_w4a16_mixtral_T{T}_weight_GBs"] = round((w1.numel() + w2.numel()) * hit / E / ms / 1e6, 1)
        out[f"fused_experts_w4a16_mixtral_T{T}_experts_hit"] = hit
    # the same layer with mxfp4 weights (e2m1 codes, one E8M0 scale byte per 32)
    s1m = torch.randint(118, 124, (E, 2 * I, Hd // 32), device=dev, dtype=torch.uint8)
    s2m = torch.randint(118, 124, (E, Hd, I // 32), device=dev, dtype=torch.uint8)
    for T in (64, 2048):
        xx = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
        logits = torch.randn(T, E, device=dev, dtype=torch.bfloat16)
        tw = torch.empty(T, topk, device=dev, dtype=torch.float32)
        ti = torch.empty(T, topk, device=dev, dtype=torch.int32)
        sgl_kernel.topk_softmax(tw, ti, logits, True)
        ms = timeit(lambda: sgl_kernel.fused_experts(xx, w1, w2, tw, ti, use_mxfp4_w4a16=True, w1_scale=s1m, w2_scale=s2m),
                    iters=40 if T <= 256 else 8)
        out[f"fused_experts_mxfp4_mixtral_T{T}_ms"] = round(ms, 4)
    out["timing"] = ("device time per call: N calls in one HIP graph, one replay timed"
                     + ("; graph capture failed, eager loop timed instead, for %d leg(s)" % len(eager_legs) if eager_legs else ""))
    return out


def roofline_extra(ex, bf16_ceiling=None):
    """Fraction of the bounding roofline per extra leg (MI355X_MICROARCH.md peaks: HBM 8 TB/s, bf16 MFMA 2.5 PFLOP/s,
    fp8 / int8 MFMA 5 P): {leg: {"bound", "achieved", "peak", "unit", "frac"}}. Latency-bound legs (routing) carry no
    fraction."""
    out = {}

    def put(leg, bound, achieved, peak, unit):
        out[leg] = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": round(achieved / peak, 4)}
        if bound == "mfma" and peak == 2500.0 and bf16_ceiling:
            # next to the nominal peak: the same-run register-only bf16 MFMA stream at two waves per SIMD (what the matrix
            # pipes deliver on this device at the clock it holds under them)
            out[leg]["ceiling_measured"] = bf16_ceiling
            out[leg]["frac_of_ceiling_measured"] = round(achieved / bf16_ceiling, 4)

    for k, v in ex.items():
        if k.endswith("_GBs") and "weight" not in k and "fp8_blockwise_gemm" not in k:
            put(k[:-4], "hbm", v, PEAK_HBM_GBS, "GB/s")
        elif k.endswith("_weight_GBs") and ("_T1_" in k or "_T32_" in k or "_T64_" in k or "_M1_" in k or "_M16_" in k or "_M64_" in k or "_M128_" in k):
            put(k[:-4], "hbm", v, PEAK_HBM_GBS, "GB/s")  # few rows: the weight stream bounds the GEMM
        elif k.endswith("_TFLOPs"):
            if k.startswith("fp8_scaled_mm"):
                put(k[:-7], "mfma", v, PEAK_FP8_TFLOPS, "TFLOP/s")
            elif "_T1_" in k or "_T32_" in k or "_T64_" in k or "flash_mla_decode" in k:
                continue  # (their bound is HBM: the GB/s entry above)
            else:
                put(k[:-7], "mfma", v, 2500.0, "TFLOP/s")
        elif k.endswith("_TOPs"):
            put(k[:-5], "mfma", v, 5000.0, "TOP/s")
    return out


def spawn_replicas(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and pass rank 0's
    JSON line through. The parent never touches the GPU (no torch.cuda call), the children are ordinary
    subprocesses (no exec from a process that has initialised the GPU)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # Poll all ranks: when one dies (bad GPU, import error) its siblings would sit in init_process_group / barrier until
    # the collective timeout, so the first non-zero exit ends the others; an overall limit ends a hang.
    import threading
    import time

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("SGLK_BENCH_TIMEOUT_S", "3000"))
    rcs = [None] * len(procs)
    failed = None
    while any(rc is None for rc in rcs):
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = (r, rcs[r])
        if failed is not None or time.time() > deadline:
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    pr.terminate()
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = pr.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        rcs[r] = pr.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    sys.stdout.write("".join(c for c in chunks if c))
    sys.stdout.flush()
    if failed is not None:
        sys.exit("bench.py: rank %d exited with %s; the other ranks were stopped" % failed)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.exit("bench.py: rank(s) failed or timed out: %s" % bad)


def init_ranks(args):
    """(rank, local_rank, world, dist or None) from the launcher's environment; --gpus must agree with it."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.plumbing_only:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    return rank, local_rank, world, dist


def timed_steps(step, steps, sync, dist, dev):
    """EXACTLY `steps` steps between barrier + sync on both sides; returns the MAX elapsed seconds over ranks."""
    def barrier():
        if dist is not None:
            dist.barrier()

    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def mla_roofline(sgl_kernel, dev, heads=128):
    """Second half of BASELINE's metric: flash_mla_decode at configs[3] (bs=128, seq=8192, kv_lora 512 + rope 64,
    paged 64, bf16, q x 100 as the reference benchmark), HIP events on the launch stream around every call.
    Bytes = reference formula benchmark/bench_flash_mla_decode.py:109-115 (q + cache + table + seq_lens + out)."""
    bs, seq, page = 128, 8192, 64
    n_pages = seq // page
    g = torch.Generator(device=dev).manual_seed(0x561)
    cache = torch.randn(bs * n_pages, page, 576, device=dev, dtype=torch.bfloat16, generator=g)
    table = torch.randint(0, bs * n_pages, (bs, n_pages), device=dev, dtype=torch.int32, generator=g)
    seq_lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
    qq = torch.randn(bs, heads, 576, device=dev, dtype=torch.bfloat16, generator=g) * 100
    q_nope, q_pe = qq[..., :512], qq[..., 512:].contiguous()
    ws = torch.empty(sgl_kernel.flash_mla_get_workspace_size(seq, bs, heads, page, -1), device=dev, dtype=torch.uint8)

    def run():
        return sgl_kernel.flash_mla_decode(q_nope, q_pe, cache, seq_lens, table, ws, 576 ** -0.5, -1)

    for _ in range(30):
        run()
    iters = 50
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b_ in ev:
        a.record()
        run()
        b_.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b_) for a, b_ in ev)
    avg_eager = sum(ms) / len(ms)
    # Device time per call = `iters` calls captured in ONE HIP graph, median of five replays (events on the launch stream: an
    # eager loop of 0.3-ms calls still carries the host's launch gaps); the eager per-call average is reported next to it.
    avg, how = avg_eager, "eager: HIP events around every call"
    try:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for _ in range(iters):
                    run()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        torch.cuda.synchronize()
        reps = []
        for _ in range(5):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            graph.replay()
            en.record()
            torch.cuda.synchronize()
            reps.append(st.elapsed_time(en) / iters)
        avg, how = sorted(reps)[2], f"{iters} calls in one HIP graph, median of 5 replays, HIP events on the launch stream"
        del graph
    except Exception:
        torch.cuda.synchronize()
    nbytes = qq.numel() * 2 + cache.numel() * 2 + table.numel() * 4 + seq_lens.numel() * 4 + bs * heads * 512 * 2
    return {
        "bound": "hbm",
        "kernel": "mla_rows128z_kernel<bf16> (32x32x16 MFMA, row per lane; the split merge inside the kernel, ONE launch)" if heads > 64 else "mla_decode_kernel<bf16> (+ mla_reduce_kernel)",
        "workload": f"flash_mla_decode bs={bs} seq={seq} heads={heads} kv_lora=512 rope=64 page={page} bf16",
        "achieved": round(nbytes / avg / 1e6, 1),
        "peak": PEAK_HBM_GBS,
        "unit": "GB/s",
        "frac": round(nbytes / avg / 1e6 / PEAK_HBM_GBS, 4),
        "bytes": nbytes,
        "traffic": mla_pmc_traffic_bytes(),
        "traffic_source": MLA_PMC_FILE + " (rocprofv3 PMC passes of this round's kernel, not this run)",
        "kernel_ms_avg": round(avg, 4),
        "timing": how,
        "eager_ms_avg": round(avg_eager, 4),
        "eager_ms_median": round(ms[len(ms) // 2], 4),
        "tflops": round(2.0 * bs * heads * seq * (576 + 512) / avg / 1e9, 1),
    }


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    # test hook (tests/test_bench_plumbing.py, CPU, gloo): rank / barrier / max-over-ranks / JSON plumbing with an
    # empty step; measures nothing and says so in the line it prints
    ap.add_argument("--plumbing-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_replicas(args, argv)
    rank, local_rank, world, dist = init_ranks(args)

    if args.plumbing_only:
        dev = torch.device("cpu")
        elapsed = timed_steps(lambda i: None, args.steps, lambda: None, dist, dev)
        if rank == 0:
            print(json.dumps({"metric": "plumbing-only (no kernels ran)", "value": None, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "elapsed_max_s": elapsed}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import sgl_kernel

    x, b, sb, q, s = make_inputs(dev, 0x561 + rank)

    def quant():
        sgl_kernel.sgl_per_token_group_quant_8bit(x, q, s, GROUP, 1e-10, -448.0, 448.0, False, enable_v2=False)

    def gemm():
        return sgl_kernel.fp8_blockwise_scaled_mm(q, b, s, sb, torch.bfloat16)

    # The part ramps its shader clock over the first tens of milliseconds of load (MI355X_MICROARCH.md, DVFS): W = 5
    # warm-up steps are 1.5 ms of work. Bring the chip to its steady state first with an untimed run of the same step
    # (reported in config.clock_ramp_ms), then do the W warm-up steps and the K timed steps as the contract says.
    # The timed region is repeated REPS times (the devices of this pool differ by ~10 % and a single 20-step sample cannot show
    # a 5 % kernel gain): every repetition is ramp + W warm-up steps + EXACTLY K timed steps between barrier + synchronize;
    # `value` / `ms_per_step` / `roofline.achieved` are the MEDIAN repetition's, all of them are listed in `repetitions`.
    flop = 2.0 * M * N * K
    reps = []
    for _ in range(REPS):
        ramp_t0 = time.perf_counter()
        while time.perf_counter() - ramp_t0 < CLOCK_RAMP_S:
            for _ in range(50):
                quant()
                gemm()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            quant()
            gemm()

        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

        def step(i):
            quant()
            ev[i][0].record()  # current stream == the stream the ops launch on
            gemm()
            ev[i][1].record()

        elapsed = timed_steps(step, args.steps, torch.cuda.synchronize, dist, dev)
        kms = sorted(a.elapsed_time(b_) for a, b_ in ev)
        reps.append({"ms_per_step": elapsed * 1e3 / args.steps, "kernel_ms_avg": sum(kms) / len(kms),
                     "kernel_ms_median": kms[len(kms) // 2], "kernel_ms_min": kms[0]})
    order = sorted(range(REPS), key=lambda r: reps[r]["ms_per_step"])
    med = reps[order[REPS // 2]]
    ms_per_step = med["ms_per_step"]
    gemm_avg_ms = sorted(r["kernel_ms_avg"] for r in reps)[REPS // 2]
    gemm_med_ms = sorted(r["kernel_ms_median"] for r in reps)[REPS // 2]
    gemm_min_ms = min(r["kernel_ms_min"] for r in reps)

    clock = gemm_clock(q, b, s, sb, dev) if rank == 0 else None
    ceiling = mfma_ceiling(dev) if rank == 0 else None

    value = world * flop / (ms_per_step * 1e-3) / 1e12
    achieved = flop / (gemm_avg_ms * 1e-3) / 1e12

    result = {
        "metric": "achieved TFLOPS (fp8 GEMM) + GB/s (flash decode) at Llama-3-8B shapes, 1 GPU",
        "value": round(value, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "fp8_e4m3",
        "data": "synthetic",
        "config": {
            "workload": "per_token_group_quant_fp8 + fp8_blockwise_scaled_mm, Llama-3-8B FFN gate/up "
                        "(M=4096, N=14336, K=4096, 1x128 / 128x128 fp32 block scales, bf16 out)",
            "M": M, "N": N, "K": K, "parallelism": "replicas" if world > 1 else "single",
            "clock_ramp_ms": int(CLOCK_RAMP_S * 1e3),
            "repetitions": REPS,
        },
        "repetitions": [{k: round(v, 4) for k, v in r.items()} | {"tflops": round(world * flop / r["ms_per_step"] / 1e9, 1)}
                        for r in reps],
        "roofline": {
            "bound": "mfma",
            "kernel": GEMM_KERNEL,
            "achieved": round(achieved, 2),
            "peak": PEAK_FP8_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP8_TFLOPS, 4),
            "traffic": pmc_traffic_bytes(),
            "traffic_source": PMC_FILE + " (rocprofv3 PMC passes of this round's kernel, not this run)",
            "kernel_ms_avg": round(gemm_avg_ms, 4),          # median over the repetitions of the per-repetition mean
            "kernel_ms_median": round(gemm_med_ms, 4),
            "kernel_ms_min": round(gemm_min_ms, 4),
            "frac_at_min": round(flop / (gemm_min_ms * 1e-3) / 1e12 / PEAK_FP8_TFLOPS, 4),
            # same process, same device: what a register-only stream of the GEMM's own MFMA reaches on random operands -
            # the clock the part holds under full matrix load sets it, not the nominal 2.4 GHz
            "ceiling_measured": ceiling,
            "ceiling_tflops": ceiling and ceiling["tflops"],  # (scalar copy: nested objects do not survive every log)
            "frac_of_ceiling_measured": ceiling and round(achieved / ceiling["tflops"], 4),
            # the kernel's own s_memtime / s_memrealtime stamps (DIAGNOSTIC build of the library, untimed launches after
            # the timed region): the shader clock the part sustains under this kernel, and the matrix-pipe share of a
            # K block in shader cycles (2 waves x 16 MFMAs x 64 cycles per SIMD and K block of a whole tile)
            "sustained_clock_mhz": clock and clock["mhz"],
            "cycles_per_k_block": clock and clock["cycles_per_k_block"],
            "mfma_busy_by_cycles": clock and round(2048.0 / clock["cycles_per_k_block"], 4),
            "half_tile_phase": clock and {k: v for k, v in clock.items() if k.startswith("half_")},
        },
    }
    if rank == 0:
        if world == 1:
            del x, b, sb, q, s
            # the "GB/s (flash decode)" half of the metric: its own roofline entry, measured in the same run
            fd = mla_roofline(sgl_kernel, dev)
            # same-run bf16 matrix ceilings (register-only 32x32x16 streams): what the 128-row MLA kernel (one wave per SIMD)
            # and the two-waves-per-SIMD kernels (fwd prefill, MoE tile pipeline) could reach on this device under load
            c4, c8 = mfma_ceiling_bf16(dev, 4), mfma_ceiling_bf16(dev, 8)
            fd["bf16_ceiling_tflops_1_wave_per_simd"] = c4
            fd["frac_of_bf16_ceiling_measured"] = c4 and round(fd["tflops"] / c4, 4)
            result["roofline_flash_decode"] = fd
            # scalar copies at the top level of `roofline` (the second half of the metric must survive a flattened log)
            result["roofline"]["flash_decode_frac"] = fd["frac"]
            result["roofline"]["flash_decode_GBs"] = fd["achieved"]
            result["roofline"]["flash_decode_ms"] = fd["kernel_ms_avg"]
            result["roofline"]["bf16_ceiling_tflops_1_wave_per_simd"] = c4
            result["roofline"]["bf16_ceiling_tflops_2_waves_per_simd"] = c8
        if world == 1 and not args.no_extra:
            result["extra"] = side_metrics(sgl_kernel, dev)
            result["roofline_extra"] = roofline_extra(result["extra"], result["roofline"].get("bf16_ceiling_tflops_2_waves_per_simd"))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
