#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c2
mkdir -p $OUT
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
{
  timeout 120 $K stamps 4096 14336 4096
  timeout 120 $K gemm 256 14336 4096 4
  timeout 120 $K gemm 1024 14336 4096 4
} > $OUT/kbench.log 2>&1
timeout 1500 python3 -m pytest tests/test_full_size_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > $OUT/pytest_new.log 2>&1
timeout 1200 python3 -m pytest tests/test_moe_gpu.py tests/test_mla_prefill_gpu.py tests/test_gemm_gpu.py -x -q -m gpu > $OUT/pytest_a.log 2>&1
timeout 600 python3 -m pytest tests/test_attention_gpu.py tests/test_mla_decode_gpu.py -x -q -m gpu -k "golden or errors or softcap or masked or full_size or other_head or late" > $OUT/pytest_b.log 2>&1
cat $OUT/kbench.log; tail -15 $OUT/pytest_new.log; tail -5 $OUT/pytest_a.log; tail -5 $OUT/pytest_b.log
