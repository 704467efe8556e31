#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
timeout 200 $K mla 128 8192 128 2>&1 | tail -3
timeout 200 $K mla 128 8192 16 2>&1 | tail -3
timeout 300 python3 tools/qserve_bench.py 128 256 4096 2>&1 | grep -v amdgpu.ids
timeout 300 python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2048 2>&1 | grep -v amdgpu.ids
for rows in 1 8,24,12,20,16,16,10,21; do
timeout 120 $K w4a16 28672 4096 $rows 0:0 0:0
timeout 120 $K w4a16 4096 14336 $rows 0:0 0:0
done
timeout 1200 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py tests/test_qserve_gpu.py tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "golden or varlen or kvcache_paged or decode_kernel" 2>&1 | tail -3
