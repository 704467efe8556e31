#!/bin/bash
# r05 lease zm: the down projection at 48 - 96 tokens (12 - 24 rows per expert): the K-split kernel forced (MOE_W4_MT=12) against the
# default tile choice, interleaved
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zm
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2; do
  for mt in 0 12; do
    echo "== MOE_W4_MT=$mt"
    MOE_W4_MT=$mt LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/moe_gemm_split.py 32 48 64 96 2>&1 | grep "^T=" | grep routed
  done
done | tee $OUT/ksplit.log
