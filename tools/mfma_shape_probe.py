"""MFMA shape probe (MI355X_MICROARCH.md, DVFS give-back item 7): register-only and LDS-fed bf16 streams on 32x32x16 and on
16x16x32, and the MX fp8 stream on 32x32x64 and 16x16x128, same FLOP and output tile per wave, random operands, one
workgroup per CU. Prints TFLOP/s per arm, arms interleaved, three rounds."""
import ctypes, os, sys, time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-xpu_amd", "build", "libsglk_ceiling.so"))
dev = torch.device("cuda:0")
blocks = torch.cuda.get_device_properties(dev).multi_processor_count
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.sglk_bench_mfma_shape_bf16.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
lib.sglk_bench_mfma_shape_fp8_16.argtypes = [vp, vp, vp, ci, ci]
lib.sglk_bench_mfma_ceiling.argtypes = [vp, vp, vp, ci, ci]
g = torch.Generator(device="cpu").manual_seed(9)
src = torch.randint(0, 2 ** 31 - 1, (16384,), generator=g, dtype=torch.int32).to(dev)
s8 = torch.randint(0, 256, (16384 * 4,), generator=g, dtype=torch.uint8)
s8[(s8 & 0x7F) == 0x7F] ^= 1
s8 = s8.to(dev)
dst = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(run, flop):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return flop / (ms[10] * 1e-3) / 1e12


for rnd in range(3):
    for waves in (4, 8):
        iters = 3000 if waves == 4 else 1500
        flop = blocks * waves * 8.0 * iters * 2.0 * 32 * 32 * 16
        row = []
        for lds in (0, 1):
            for shape in (0, 1):
                t = timed(lambda: lib.sglk_bench_mfma_shape_bf16(st, src.data_ptr(), dst.data_ptr(), blocks, waves, iters, shape, lds), flop)
                row.append("%s%s %.0f" % ("32x32x16" if shape == 0 else "16x16x32", " lds" if lds else "", t))
        print("bf16 waves/WG %d: " % waves + " | ".join(row), flush=True)
    iters = 1500
    flop = blocks * 8.0 * 8.0 * iters * 2.0 * 32 * 32 * 64
    t0 = timed(lambda: lib.sglk_bench_mfma_ceiling(st, s8.data_ptr(), dst.data_ptr(), blocks, iters), flop)
    t1 = timed(lambda: lib.sglk_bench_mfma_shape_fp8_16(st, s8.data_ptr(), dst.data_ptr(), blocks, iters), flop)
    print("fp8 MX 8 waves/WG: 32x32x64 %.0f | 16x16x128 %.0f" % (t0, t1), flush=True)
