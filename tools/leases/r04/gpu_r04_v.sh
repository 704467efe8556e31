#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_v
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 2400 python3 -m pytest tests/test_attention_gpu.py tests/test_gemm_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
SPLITS=0,2,4 timeout 300 python3 tools/attn_decode_sweep.py 2>&1 | grep -v amdgpu > $OUT/sweep.log
cat $OUT/sweep.log
