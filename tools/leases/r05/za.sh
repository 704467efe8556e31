#!/bin/bash
# r05 lease za: gate / up at 128 rows per expert: 128 x 512 tiles (WIDE) against 128 x 256 tiles (MS = 2), one box, interleaved
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_za
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2; do
  for w in 1 0; do
    echo "== MOE_WIDE=$w"
    MOE_WIDE=$w LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/moe_gemm_split.py 384 512 640 2>&1 | grep "^T="
  done
done | tee $OUT/wide_ab.log
