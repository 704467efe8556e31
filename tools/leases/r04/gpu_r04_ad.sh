#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_ad
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
echo "== release (4 waves, DMA)"
timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu | tee $OUT/prefill.log
timeout 1500 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -n 4 2>&1 | tail -4
for rep in 1 2; do
for w in 8 4; do
  echo "== waves $w"
  ATTN_PREFILL_WAVES=$w LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done
done | tee -a $OUT/prefill.log
for pr in 1 6 7 8 16 32 40 63 127 191 255; do
  echo "== waves 4 probe $pr"
  ATTN_PREFILL_ONLY=128 ATTN_PREFILL_WAVES=4 ATTN_PREFILL_PROBE=$pr LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done | tee -a $OUT/prefill.log
