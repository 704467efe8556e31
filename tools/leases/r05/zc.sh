#!/bin/bash
# r05 lease zc: QServe W4A8 decode, 17 - 64 rows: the 32x32x32 form (128 columns per wave, 32-row m-tiles, half the workgroups)
# against the default stream, bit-identity checked (tools/qserve_bench.py QSERVE_CFGS)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zc
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
QSERVE_CFGS=320022,320042,320044,320082,320084,310042,310044,310084 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 600 python3 tools/qserve_bench.py 17 32 48 64 2>&1 | grep -v amdgpu | tee $OUT/qserve.log
