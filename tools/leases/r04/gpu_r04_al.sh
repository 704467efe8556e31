#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_al
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu | tee $OUT/prefill.log
timeout 1500 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py tests/test_graph_capture_gpu.py tests/test_determinism_gpu.py -m gpu -q -x -n 4 2>&1 | tail -3
