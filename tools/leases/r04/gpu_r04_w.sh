#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_w
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for mf in 0 1 2; do
  echo "== QSERVE_MF=$mf"
  QSERVE_MF=$mf LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/qserve_bench.py 32 64 2>&1 | grep -v amdgpu
done > $OUT/qserve.log 2>&1
cat $OUT/qserve.log
