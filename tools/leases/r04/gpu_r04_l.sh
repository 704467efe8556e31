#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_l
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log; grep -n "^FAILED" $OUT/pytest.log | head -20
timeout 600 python3 tools/moe_bench.py 256 384 512 768 1024 > $OUT/moe.log 2>&1
head -6 $OUT/moe.log
