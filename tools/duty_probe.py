#!/usr/bin/env python3
"""Is the headline GEMM limited by instantaneous current / power or by power averaged over a longer window?

Runs fp8_blockwise_scaled_mm (4096, 14336, 4096) back to back and with idle gaps of growing length between launches
(one spinning lane, everything else idle) and prints the GEMM's own duration (HIP events around each launch). If the
GEMM gets faster as the duty cycle falls, the part budgets power over a window longer than a launch: idle bubbles
inside a launch are then refunded as clock, and only energy per GEMM buys throughput. If it stays put, every idle
cycle inside the launch is lost time.

    python tools/duty_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sgl-kernel-xpu_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
import sgl_kernel  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    x, b, sb, q, s = bench.make_inputs(dev, 0x561)
    sgl_kernel.sgl_per_token_group_quant_8bit(x, q, s, 128, 1e-10, -448.0, 448.0, False, enable_v2=False)

    def gemm():
        return sgl_kernel.fp8_blockwise_scaled_mm(q, b, s, sb, torch.bfloat16)

    # calibrate torch.cuda._sleep
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(10_000_000)
    e1.record()
    torch.cuda.synchronize()
    cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
    print(f"_sleep: {cyc_per_us:.1f} cycles per us")

    for gap_us in (0, 25, 50, 100, 200, 400, 800, 1600, 0):
        n = 400 if gap_us <= 200 else 150
        # ramp in this regime
        for _ in range(n):
            gemm()
            if gap_us:
                torch.cuda._sleep(int(gap_us * cyc_per_us))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b_ in ev:
            a.record()
            gemm()
            b_.record()
            if gap_us:
                torch.cuda._sleep(int(gap_us * cyc_per_us))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b_) for a, b_ in ev)
        print(f"gap {gap_us:5d} us: gemm median {ms[len(ms) // 2] * 1e3:7.1f} us  min {ms[0] * 1e3:7.1f}  "
              f"p90 {ms[int(len(ms) * 0.9)] * 1e3:7.1f}", flush=True)


if __name__ == "__main__":
    main()
