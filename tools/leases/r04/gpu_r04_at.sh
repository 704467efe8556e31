#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_at
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2; do
for pr in 0 1024; do
  echo "== setprio probe $pr"
  ATTN_PREFILL_PROBE=$pr LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done
done | tee $OUT/prefill_prio.log
