#!/bin/bash
# round 4, call A: power-management probe, MFMA ceilings, baseline bench on this box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_a
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 300 python3 tools/duty_probe.py > $OUT/duty.log 2>&1
{
  timeout 120 $K peak 512 256 4000
  timeout 120 $K peak 256 256 4000
  GEMM_CLOCK=1 timeout 200 $K gemm 4096 14336 4096 4 22
} > $OUT/kbench.log 2>&1
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/duty.log $OUT/kbench.log; cat $OUT/bench.json; tail -3 $OUT/bench.err
