#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_ax
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2; do
for w in 4 64; do
  echo "== waves $w"
  ATTN_PREFILL_WAVES=$w LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu | grep "d=64"
done
done | tee $OUT/prefill_mb64.log
