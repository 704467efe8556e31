#!/bin/bash
# r05 lease zk: flash_mla_decode H = 128, two splits of unequal length (probe 9: the first split one tile shorter, so that it
# publishes while the merger still streams) against equal ones, interleaved on one box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zk
mkdir -p $OUT
KB=$R/sgl-kernel-xpu_amd/build/kbench
for rep in 1 2 3 4; do
  echo "== equal";   MLA_GAUSS=100 timeout 120 $KB mla 128 8192 128 2>&1 | grep -i "ms\|TB" | head -2
  echo "== uneven";  MLA_GAUSS=100 MLA_PROBE=9 timeout 120 $KB mla 128 8192 128 2>&1 | grep -i "ms\|TB" | head -2
done | tee $OUT/ab.log
