#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_p
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 300 python3 tools/attn_decode_sweep.py > $OUT/sweep.log 2>&1
grep "splits= [0124]:" $OUT/sweep.log
