#!/bin/bash
# r05 lease z: where fused_experts T = 512 spends its time: the two W4A16 grouped GEMMs one by one (uniform / routed rows), the whole layer per launch
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_z
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 600 python3 tools/moe_gemm_split.py 256 512 1024 2>&1 | grep -v amdgpu | tee $OUT/split.log
cd /tmp && export TMPDIR=/tmp
MOE_BENCH_INT4_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o moe512 -- python3 $R/tools/moe_bench.py 512 > $OUT/moe512.log 2>&1
tail -3 $OUT/moe512.log
f=$(ls $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
