#!/bin/bash
# r05 lease zq: QServe W4A8 over a deep K (N = 4096, K = 14336; 8192^2) across rows: parity + timing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zq
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_qserve_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
QSERVE_SHAPES="4096x14336,8192x8192,4096x4096,14336x4096" timeout 600 python3 tools/qserve_bench.py 64 128 129 192 256 257 384 512 513 768 1024 1025 2048 2>&1 | grep "N=" | tee $OUT/qserve.log
