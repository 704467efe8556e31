#!/bin/bash
# r05 lease g: MLA wall time of the un-stamped probe variants (what each ingredient costs in TIME) and of the ring depths
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_g
mkdir -p $OUT
cd $R/sgl-kernel-xpu_amd/build
{
MLA_GAUSS=100 MLA_TIME_VARIANTS=0,204,203,205,206 timeout 300 ./kbench mla 128 8192 128
MLA_GAUSS=100 MLA_TIME_VARIANTS=0,270,271,272,273,274,276 timeout 300 ./kbench mla 128 8192 128
} 2>&1 | tee $OUT/mla_variants.log
