#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
K=$R/sgl-kernel-xpu_amd/build/kbench
timeout 120 $K stream 229376 2048 2>&1 | grep "pattern=3\|pattern=0 depth=2"
timeout 120 $K stream 32768 7168 2>&1 | grep "pattern=3\|pattern=0 depth=2"
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 300 python3 tools/qserve_bench.py 2>&1 | grep -v amdgpu.ids
