#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_ab
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for pr in 0 1 2 4 6 8 16 32 40 7 63; do
  echo "== probe $pr"
  ATTN_PREFILL_ONLY=128 ATTN_PREFILL_PROBE=$pr LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done | tee $OUT/prefill_probes.log
