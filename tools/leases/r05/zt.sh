#!/bin/bash
# r05 lease zt: flash_mla_decode with more than 64 heads and more than four splits on the 8-wave kernel + parallel reduce: parity, the
# (H, batch, length) sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zt
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python3 tools/row_sweep.py mla 2>&1 | grep "flash_mla" | grep -E "H=(64|65|96|128) " | tee $OUT/sweep.log
