#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_s
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 900 python3 -m pytest tests/test_gemm_gpu.py tests/test_quant_extra_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
{
for m in 8 16; do
  GEMM_LA_ROWS=99 timeout 60 $K gemm $m 14336 4096 4
  GEMM_LA_ROWS=8 timeout 60 $K gemm $m 14336 4096 4
  GEMM_LA_ROWS=99 timeout 60 $K gemm $m 4096 14336 4
  GEMM_LA_ROWS=8 timeout 60 $K gemm $m 4096 14336 4
done
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
