#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_n
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log; grep -n "^FAILED" $OUT/pytest.log | head
timeout 600 python3 tools/moe_bench.py 512 > $OUT/moe.log 2>&1
cat $OUT/moe.log
