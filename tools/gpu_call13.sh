#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c13
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 900 python3 -m pytest tests/test_moe_gpu.py -q -m gpu -k "test_moe_grouped_mm_w4a16" 2>&1 | grep -v "^E \|^tests\|^$\|^    \|^>" | cut -c1-150 > $OUT/pytest_w4.log
grep -c FAILED $OUT/pytest_w4.log; grep "FAILED\|passed\|failed" $OUT/pytest_w4.log | head -80
timeout 300 python3 tools/attn_bench.py > $OUT/attn_bench.log 2>&1
cat $OUT/attn_bench.log
