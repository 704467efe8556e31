#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c4
mkdir -p $OUT
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
{
  timeout 120 $K issue 512 256 1000
  timeout 120 $K issue 256 256 1000
  timeout 300 $K gemm 4096 14336 4096 4 20 21 4 20 21
} > $OUT/kbench.log 2>&1
timeout 300 python3 tools/power_probe.py > $OUT/power.log 2>&1
timeout 1500 python3 -m pytest tests/test_attn_aux_gpu.py -x -q -m gpu > $OUT/pytest_aux.log 2>&1
timeout 900 python3 -m pytest tests/test_activation_gpu.py tests/test_moe_gpu.py -x -q -m gpu -k "golden" > $OUT/pytest_gold.log 2>&1
cat $OUT/kbench.log; cat $OUT/power.log; tail -15 $OUT/pytest_aux.log; tail -5 $OUT/pytest_gold.log
