#!/bin/bash
# r05 lease zr: MoE tile-pipeline thresholds moved (88 rows per expert for 128-row blocks, 152 for 256-row blocks): parity of the MoE
# files, the token sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zr
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py tests/test_determinism_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python3 tools/row_sweep.py moe 2>&1 | grep "T=" | tee $OUT/sweep.log
