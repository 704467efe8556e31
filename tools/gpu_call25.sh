#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 1500 python3 -m pytest tests/test_gemm_gpu.py tests/test_qserve_gpu.py tests/test_moe_gpu.py tests/test_determinism_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2048 2>&1 | grep -v amdgpu.ids
timeout 300 python3 tools/qserve_bench.py 2>&1 | grep -v amdgpu.ids
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2
