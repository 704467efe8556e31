#!/bin/bash
# round 4, call C: the one-launch fp8 blockwise GEMM - parity, then stagger policies A/B, in-kernel clocks
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_c
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 900 python3 -m pytest tests/test_gemm_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
{
  ROUNDS=6 timeout 300 $K gemmab 4096 14336 4096 4:0 4:1 4:2
  timeout 100 $K gemm 256 14336 4096 4
  timeout 100 $K gemm 512 14336 4096 4
  timeout 100 $K gemm 1024 4096 4096 4
  GEMM_CLOCK=1 GEMM_STAGGER=0 timeout 100 $K gemm 4096 14336 4096 4
  GEMM_CLOCK=1 GEMM_STAGGER=1 timeout 100 $K gemm 4096 14336 4096 4
  GEMM_CLOCK=1 GEMM_STAGGER=2 timeout 100 $K gemm 4096 14336 4096 4
  ROUNDS=4 timeout 300 $K gemmab 8192 4096 14336 4:0 4:1 4:2
  ROUNDS=4 timeout 300 $K gemmab 2048 14336 4096 4:0 4:1
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
