#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
K=$R/sgl-kernel-xpu_amd/build/kbench
timeout 120 $K w4a16 28672 4096 1 0:1 16:1 24:1 48:1 80:1 112:1 120:1 0:1
timeout 120 $K w4a16 4096 14336 1 0:1 16:1 24:1 48:1 80:1 112:1 120:1 0:1
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 300 python3 tools/qserve_bench.py 2>&1 | grep -v amdgpu.ids
timeout 900 python3 -m pytest tests/test_attention_gpu.py -x -q -m gpu -k "decode_kernel_features" 2>&1 | tail -3
