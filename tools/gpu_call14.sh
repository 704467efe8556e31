#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c14
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
timeout 900 python3 -m pytest tests/test_moe_gpu.py -q -m gpu -k "w4a16 or golden or mxfp4 or mixtral" 2>&1 | tail -5
{
for r in 1 4 16; do
  timeout 120 $K w4a16 28672 4096 $r 0:0 1:0 2:0 4:0 8:0 16:0 0:2 1:2
  timeout 120 $K w4a16 4096 14336 $r 0:0 1:0 2:0 4:0 8:0 16:0 0:2 1:2
done
} > $OUT/kbench_w4.log 2>&1
cat $OUT/kbench_w4.log
timeout 300 python3 tools/attn_bench.py > $OUT/attn_bench.log 2>&1
cat $OUT/attn_bench.log
timeout 1200 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "golden or full or prefill or causal" 2>&1 | tail -5
