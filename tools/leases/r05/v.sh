#!/bin/bash
# r05 lease v: kQ16 (216) against the 32-wide loop (232) over decode shapes
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_v
mkdir -p $OUT
cd $R/sgl-kernel-xpu_amd/build
{
for shape in "32 8192 128" "64 4096 128" "256 2048 128" "64 1024 128" "128 8192 96" "16 8192 128" "128 1024 128"; do
MLA_GAUSS=100 MLA_TIME_VARIANTS=216,232 timeout 300 ./kbench mla $shape | tail -4
done
} 2>&1 | tee $OUT/mla.log
