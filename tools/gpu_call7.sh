#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c7
mkdir -p $OUT
cd $R
timeout 1800 python3 -m pytest tests/test_sampling_gpu.py -x -q -m gpu > $OUT/pytest_sampling.log 2>&1
tail -25 $OUT/pytest_sampling.log
