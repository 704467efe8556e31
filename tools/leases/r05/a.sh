#!/bin/bash
# r05 lease a: baseline of the MLA H=128 decode on this round's first box: wall time, in-kernel stamps of the release loop and
# of the probes that drop one ingredient each (garbage results)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_a
mkdir -p $OUT
cd $R/sgl-kernel-xpu_amd/build
export LD_LIBRARY_PATH=$PWD:$LD_LIBRARY_PATH
{
MLA_GAUSS=100 timeout 120 ./kbench mla 128 8192 128
MLA_GAUSS=100 MLA_STAMPS=0,1,2,3,4,5,6,12,16 timeout 300 ./kbench mla 128 8192 128
MLA_GAUSS=100 timeout 120 ./kbench mla 128 8192 128
timeout 120 ./kbench peak 256 1024 4000 2>&1 | tail -6
} 2>&1 | tee $OUT/mla_base.log
