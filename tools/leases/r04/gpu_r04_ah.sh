#!/bin/bash
# round 4, call AH: refreshed profile sets for the kernels that changed late in the round (prefill, fwd decode, moe, bench)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash tools/gpu_profiles_r04.sh "prefill attn moe bench"
tail -5 $R/gpurun_out/r04/prof_all.log
