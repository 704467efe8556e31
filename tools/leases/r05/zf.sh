#!/bin/bash
# r05 lease zf: what each part of the fp8 block-scale K block costs under the power cap (time, in-kernel clock, cycles per K block):
# full (4), no promotion FMAs (39), no LDS fragment reads (40), neither (41), no DMA (23), no stores (25), DMA that always hits
# L2 (26), the 16 x 16 x 128 two-launch form (22); random data, all 256 CUs, garbage results for the probes
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zf
mkdir -p $OUT
KB=$R/sgl-kernel-xpu_amd/build/kbench
for rep in 1 2; do
  GEMM_CLOCK=1 timeout 600 $KB gemm 4096 14336 4096 4 39 40 41 23 25 26 22 2>&1
done | tee $OUT/energy.log
