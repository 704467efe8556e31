#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c3
mkdir -p $OUT
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 900 python3 -m pytest tests/test_gemm_gpu.py -x -q -m gpu > $OUT/pytest_gemm.log 2>&1
{
  timeout 300 $K gemm 4096 14336 4096 4 14 15 16 17
  timeout 120 $K stamps 4096 14336 4096
  timeout 120 $K gemm 256 14336 4096 4
  timeout 120 $K gemm 1024 14336 4096 4
  timeout 120 $K gemm 8192 8192 8192 4
} > $OUT/kbench.log 2>&1
timeout 1500 python3 -m pytest tests/test_full_size_gpu.py tests/test_determinism_gpu.py -q -m gpu > $OUT/pytest_new.log 2>&1
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/pytest_gemm.log; head -8 $OUT/kbench.log; tail -6 $OUT/kbench.log; tail -15 $OUT/pytest_new.log; cat $OUT/bench.json; tail -3 $OUT/bench.err
