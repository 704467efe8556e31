#!/bin/bash
# r05 lease s: flash_mla_decode H = 128 with QK^T on v_mfma_f32_16x16x32 (kQ16): parity, then wall time next to the 32-wide form
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_s
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
MLA_GAUSS=100 MLA_TIME_VARIANTS=216,232,216,232 timeout 300 ./kbench mla 128 8192 128
MLA_GAUSS=100 MLA_TIME_VARIANTS=216,232 timeout 300 ./kbench mla 32 8192 128
} 2>&1 | tee $OUT/mla.log
