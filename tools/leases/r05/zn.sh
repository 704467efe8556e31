#!/bin/bash
# r05 lease zn: QServe W4A8 at 65 - 512 rows on the 32x32x32 stream kernel where its estimate beats the other paths: parity, timing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zn
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_qserve_gpu.py tests/test_determinism_gpu.py tests/test_cabi.py -m gpu -q -k "qserve or cabi or golden or per_" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python3 tools/qserve_bench.py 64 65 96 128 192 256 384 512 1024 2>&1 | grep "N=" | tee $OUT/qserve.log
