#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
K=$R/sgl-kernel-xpu_amd/build/kbench
for rows in 8,24,12,20,16,16,10,21 14,18,15,17,16,16,13,19 2,6,3,5,4,4,2,6 4,12,6,10,8,8,5,11 20,44,24,40,32,32,21,43; do
timeout 120 $K w4a16 28672 4096 $rows 0:1 0:1 0:2 0:2 0:4
timeout 120 $K w4a16 4096 14336 $rows 0:1 0:1 0:2 0:2 0:4
done
