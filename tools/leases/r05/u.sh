#!/bin/bash
# r05 lease u: what P.V on the 16-wide MFMA shape would buy (timing probe 208, garbage results) next to kQ16 (216) and the 32-wide loop (232)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_u
mkdir -p $OUT
cd $R/sgl-kernel-xpu_amd/build
MLA_GAUSS=100 MLA_TIME_VARIANTS=208,216,232 timeout 300 ./kbench mla 128 8192 128 2>&1 | tee $OUT/mla.log
