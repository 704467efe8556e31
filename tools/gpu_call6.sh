#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c6
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gates_gpu.py -x -q -m gpu > $OUT/pytest_gates.log 2>&1
tail -15 $OUT/pytest_gates.log
