#!/bin/bash
# r05 lease zd: QServe 33 - 64 rows at wide N on the 32x32x32 form by default: parity, then the old stream (cfg 1) and shallower rings beside it
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zd
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_qserve_gpu.py tests/test_cabi.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
QSERVE_CFGS=1,310011,311011,311021,311022 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 600 python3 tools/qserve_bench.py 41 48 64 2>&1 | grep -v amdgpu | grep -A6 "N=14336" | tee $OUT/qserve.log
