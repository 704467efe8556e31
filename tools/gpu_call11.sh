#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c11
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 300 python3 tools/attn_bench.py > $OUT/attn_bench.log 2>&1
timeout 600 python3 tools/attn_decode_sweep.py > $OUT/decode_sweep.log 2>&1
cat $OUT/attn_bench.log $OUT/decode_sweep.log
timeout 2400 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py tests/test_determinism_gpu.py -x -q -m gpu -k "not mixtral" > $OUT/pytest_attn.log 2>&1
tail -15 $OUT/pytest_attn.log
