#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_ba
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2 3; do
echo "== new (three workgroups per CU at d = 64)"
timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
echo "== old (committed)"
LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_old.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done | tee $OUT/prefill_mask.log
