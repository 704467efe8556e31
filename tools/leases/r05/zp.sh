#!/bin/bash
# r05 lease zp: fp8 block-scale GEMM: the boundary between the weight stream and the tile pipeline moved (72 rows at wide N, 256 rows
# at narrow N): parity of the GEMM file, the row sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zp
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_gemm_gpu.py tests/test_determinism_gpu.py -m gpu -q -k "gemm or blockwise or fp8 or int8" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python3 tools/row_sweep.py gemm 2>&1 | grep "N=" | tee $OUT/sweep.log
