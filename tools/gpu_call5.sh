#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c5
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_attn_aux_gpu.py -q -m gpu > $OUT/pytest_aux.log 2>&1
timeout 1200 python3 -m pytest tests/test_moe_gpu.py tests/test_activation_gpu.py tests/test_full_size_gpu.py -q -m gpu > $OUT/pytest_moe.log 2>&1
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -8 $OUT/pytest_aux.log; tail -8 $OUT/pytest_moe.log; cat $OUT/bench.json; tail -3 $OUT/bench.err
