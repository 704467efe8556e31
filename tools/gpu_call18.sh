#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c18
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
{
for r in 1 16 32; do
  timeout 120 $K w4a16 28672 4096 $r 0:0 16:0
  timeout 120 $K w4a16 4096 14336 $r 0:0 16:0
done
} > $OUT/kbench_w4.log 2>&1
cat $OUT/kbench_w4.log
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2048 2>&1 | grep -v amdgpu.ids
timeout 300 python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
timeout 300 python3 tools/attn_decode_sweep.py 2>&1 | grep "random"
timeout 900 python3 -m pytest tests/test_moe_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 2400 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py tests/test_determinism_gpu.py -x -q -m gpu -k "not mixtral" 2>&1 | tail -5
