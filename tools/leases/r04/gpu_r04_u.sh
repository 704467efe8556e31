#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_u
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 300 python3 tools/attn_layout_probe.py 2>&1 | grep -v amdgpu > $OUT/layout.log
cat $OUT/layout.log
for m in 2 4; do
  GEMM_LA_ROWS=99 timeout 60 $K gemm $m 14336 4096 4
  GEMM_LA_ROWS=2 timeout 60 $K gemm $m 14336 4096 4
done > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
