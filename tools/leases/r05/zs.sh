#!/bin/bash
# r05 lease zs: flash_mla_decode H = 128 at small batches: the rows128z kernel (in-kernel merge by ONE workgroup per batch element)
# against the 8-wave 16-head-group kernel + the parallel reduce launch (MLA_W=1), by batch size
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zs
mkdir -p $OUT
KB=$R/sgl-kernel-xpu_amd/build/kbench
for bs in 1 4 16 32 64 128; do
  for w in 0 1; do
    if [ $w = 1 ]; then export MLA_W=1; else unset MLA_W; fi
    echo -n "bs=$bs MLA_W=$w: "; MLA_GAUSS=100 timeout 120 $KB mla $bs 8192 128 2>&1 | grep "median" | head -1
  done
done | tee $OUT/ab.log
for bs in 16 128 256; do
  for w in 0 1; do
    if [ $w = 1 ]; then export MLA_W=1; else unset MLA_W; fi
    echo -n "bs=$bs seq=1024 MLA_W=$w: "; MLA_GAUSS=100 timeout 120 $KB mla $bs 1024 128 2>&1 | grep "median" | head -1
  done
done | tee -a $OUT/ab.log
