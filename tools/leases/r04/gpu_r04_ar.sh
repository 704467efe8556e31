#!/bin/bash
# round 4, call AR: refreshed profile set for the final prefill kernel
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash tools/gpu_profiles_r04.sh "prefill"
tail -30 $R/gpurun_out/r04/prof/attn_prefill.summary.txt
