#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_j
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py -m gpu -q -k "bias_enters or test_moe_grouped_mm_w4a16" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log; grep -n "AssertionError: (" $OUT/pytest.log | head -5
