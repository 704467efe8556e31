#!/bin/bash
# r05 lease t: kQ16 with the swap hazard fixed: parity (no -x), stamps of the 16-wide / 32-wide QK^T builds, wall time
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_t
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
MLA_GAUSS=100 MLA_STAMPS=86,87 timeout 300 ./kbench mla 128 8192 128
MLA_GAUSS=100 MLA_TIME_VARIANTS=216,232 timeout 300 ./kbench mla 128 8192 128
} 2>&1 | tee $OUT/mla.log
