#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c19
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
{
for r in 1 4 16; do
  timeout 120 $K w4a16 28672 4096 $r 0:0 16:0 0:1 0:2
  timeout 120 $K w4a16 4096 14336 $r 0:0 16:0 0:1 0:2
done
} > $OUT/kbench_w4.log 2>&1
cat $OUT/kbench_w4.log
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2048 2>&1 | grep -v amdgpu.ids
timeout 300 python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
timeout 600 python3 -m pytest tests/test_qserve_gpu.py -q -m gpu -x 2>&1 | tail -3
