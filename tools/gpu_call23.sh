#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
K=$R/sgl-kernel-xpu_amd/build/kbench
for r in 1 16; do
timeout 120 $K w4a16 28672 4096 $r 0:1 0:1 16:1 112:1 0:2 0:2
timeout 120 $K w4a16 4096 14336 $r 0:1 0:1 16:1 112:1 0:2 0:2
done
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2>&1 | grep -v amdgpu.ids
timeout 900 python3 -m pytest tests/test_moe_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python3 tools/qserve_bench.py 1 16 32 64 2>&1 | grep -v amdgpu.ids
timeout 600 python3 -m pytest tests/test_qserve_gpu.py -q -m gpu -x 2>&1 | tail -3
