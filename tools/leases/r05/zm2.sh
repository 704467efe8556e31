#!/bin/bash
# r05 lease zm2: kernel trace of fused_experts int4 at a DeepSeek-V3-like shape, 1 and 16 tokens (which launches carry the time?)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PROF_ROUND=r05
for t in 1 16; do
  MOE_TS=$t MOE_SHAPE=256,7168,2048,8 tools/gpu_prof.sh moe_dsv3_T$t python3 $R/tools/row_sweep.py moe > /dev/null 2>&1
  echo "== T=$t"; head -12 $R/gpurun_out/r05/prof/digest/moe_dsv3_T${t}_kernel_stats.csv | cut -c1-150
done
