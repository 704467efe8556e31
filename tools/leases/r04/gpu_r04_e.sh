#!/bin/bash
# round 4, call E: flash_mla_decode (split merge in the kernel) + the QServe W4A8 decode stream kernel: parity, timing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_e
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 1500 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_qserve_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_cabi.py -m gpu -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
{
  MLA_GAUSS=100 timeout 200 $K mla 128 8192 128 -1 2
  MLA_GAUSS=100 timeout 200 $K mla 128 8192 64 -1
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
timeout 300 python3 tools/qserve_bench.py 1 16 32 64 128 > $OUT/qserve.log 2>&1
cat $OUT/qserve.log
