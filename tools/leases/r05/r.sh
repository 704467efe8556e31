#!/bin/bash
# r05 lease r: MFMA shape probe - what the matrix pipes deliver on 16x16x32 / 16x16x128 next to 32x32x16 / 32x32x64 (same FLOP
# and output tile per wave, random operands, register-only and LDS-fed)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_r
mkdir -p $OUT
cd $R
timeout 300 python3 tools/mfma_shape_probe.py 2>&1 | grep -v amdgpu | tee $OUT/shape.log
