#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c8
mkdir -p $OUT
cd $R
timeout 1800 python3 -m pytest tests/test_attention_gpu.py -x -q -m gpu -k "kvcache or golden or errors or decode_full or fp8" > $OUT/pytest_attn.log 2>&1
tail -25 $OUT/pytest_attn.log
