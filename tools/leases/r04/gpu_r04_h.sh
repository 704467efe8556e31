#!/bin/bash
# round 4, call H: MoE bias on the tile pipeline (parity), GEMM stagger by XCD
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_h
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
{
  ROUNDS=6 timeout 300 $K gemmab 4096 14336 4096 4:0 4:3 4:4 4:1
  GEMM_CLOCK=1 GEMM_STAGGER=3 timeout 100 $K gemm 4096 14336 4096 4
  GEMM_CLOCK=1 GEMM_STAGGER=4 timeout 100 $K gemm 4096 14336 4096 4
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
