#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_o
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 2000 python3 -m pytest tests/test_attention_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_cabi.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 300 python3 tools/attn_decode_sweep.py > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
