#!/bin/bash
# r05 lease zz: moe_align_block_size with multi-workgroup counting: parity + the small-op sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zz
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py tests/test_graph_capture_gpu.py -m gpu -q -k "align or capture or graph" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 600 python3 tools/row_sweep.py elem3 2>&1 | grep "E=" | tee $OUT/elem3.log
