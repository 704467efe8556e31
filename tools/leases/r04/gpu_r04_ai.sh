#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_ai
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for w in 4 64 4 64; do
  echo "== waves $w"
  ATTN_PREFILL_WAVES=$w LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done | tee $OUT/prefill_mb.log
ATTN_PREFILL_WAVES=64 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 900 python3 -m pytest tests/test_attention_gpu.py -m gpu -q -x -n 4 2>&1 | tail -3
