#!/bin/bash
# r05 lease zw: qk-norm + rope (32-bit row division, 16-byte angle loads, sincosf) and the grouped router's LDS group scan:
# parity of the elementwise / router files + the small-op sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zw
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -k "rope or gate or topk or router or qknorm or qk_norm" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python3 tools/row_sweep.py elem3 2>&1 | grep "tokens=\|E=256" | tee $OUT/elem3.log
timeout 600 python3 tools/row_sweep.py gemm 2>&1 | grep "N=" | tee $OUT/gemm.log
