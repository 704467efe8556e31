#!/bin/bash
# r05 lease zl: the streaming kernels' gpt-oss swiglu on the plain tile layout (lane-pair epilogue): parity, GEMM 1 fused against
# GEMM + swiglu op at decode sizes
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zl
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_moe_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 300 python3 tools/gpt_oss_swiglu_bench.py 1 16 64 128 256 2>&1 | grep "T=" | tee $OUT/bench.log
