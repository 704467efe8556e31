#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_r
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for pr in 0 4 0 4; do
  echo "== MOE_PRIO=$pr"
  MOE_PRIO=$pr LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/moe_gemm_split.py 2048 2>&1 | grep -v "^   " | tail -2
done > $OUT/prio.log 2>&1
cat $OUT/prio.log
