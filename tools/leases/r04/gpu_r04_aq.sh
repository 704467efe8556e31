#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_aq
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2 3; do
echo "== new"; timeout 300 python3 tools/gemm_ab.py 2>&1 | grep -v amdgpu
echo "== old"; LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_old.so timeout 300 python3 tools/gemm_ab.py 2>&1 | grep -v amdgpu
done | tee $OUT/gemm_ab.log
timeout 1200 python3 -m pytest tests/test_gemm_gpu.py -m gpu -q -x -n 4 -k "blockwise" 2>&1 | tail -3
