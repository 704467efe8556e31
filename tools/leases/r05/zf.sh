#!/bin/bash
# r05 lease zf: fwd across head layouts and page sizes
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zf
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 tools/row_sweep.py fwdcfg 2>&1 | grep "fwdcfg" | tee $OUT/fwdcfg.log
