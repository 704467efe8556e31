#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c15
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
{
timeout 120 $K stream 229376 2048
timeout 120 $K stream 32768 7168
} > $OUT/kbench_stream.log 2>&1
cat $OUT/kbench_stream.log
for pr in 0 1 2 3; do
  echo "SGLK_ATTN_PROBE=$pr"
  SGLK_ATTN_PROBE=$pr timeout 300 python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
timeout 1200 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "golden or full or prefill or causal or decode" 2>&1 | tail -3
