#!/bin/bash
# r05 lease h: QServe W4A8 stream kernel (16-byte activation loads, split ring depths, parallel epilogue): parity + sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_h
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_qserve_gpu.py tests/test_cabi.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so QSERVE_CFGS=43082,43042,43022,43084,42082,23084,23044,23082,24044,13088,13084,14044,14042 timeout 900 python3 tools/qserve_bench.py 1 16 32 48 64 2>&1 | grep -v amdgpu | tee $OUT/qserve.log
