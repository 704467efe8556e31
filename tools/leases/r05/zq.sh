#!/bin/bash
# r05 lease zq: fused_experts at 130 - 190 rows per expert: 256-row blocks from a lower average (MOE_MIN_ROWS) against the default 192
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zq
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2; do
  for mr in 192 160 136; do
    echo "== MOE_MIN_ROWS=$mr"
    MOE_MIN_ROWS=$mr MOE_BENCH_INT4_ONLY=1 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/moe_bench.py 512 544 576 640 704 767 2>&1 | grep "fused_experts T"
  done
done | tee $OUT/minrows.log
