#!/bin/bash
# r05 lease zg: flash_mla_decode across page sizes
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zg
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 tools/row_sweep.py mlacfg 2>&1 | grep "mlacfg" | tee $OUT/mlacfg.log
