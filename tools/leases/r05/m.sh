#!/bin/bash
# r05 lease m: fp8 blockwise GEMM with the whole-tile stores staged through LDS (variant 4 = release) against round 4's direct
# row-per-lane stores (variant 37): parity, then interleaved timing on one box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_m
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_gemm_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
ROUNDS=6 timeout 300 ./kbench gemmab 4096 14336 4096 4:0 37:0
ROUNDS=4 timeout 300 ./kbench gemmab 4096 4096 14336 4:0 37:0
ROUNDS=4 timeout 300 ./kbench gemmab 8192 8192 8192 4:0 37:0
} 2>&1 | tee $OUT/gemm_ab.log
