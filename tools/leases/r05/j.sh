#!/bin/bash
# r05 lease j: MLA DMA with the nt cache policy (probe 207) against the default; fwd decode after the fp8 V-staging fix;
# QServe defaults after the sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_j
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 600 python3 -m pytest tests/test_attention_gpu.py tests/test_qserve_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
MLA_GAUSS=100 MLA_TIME_VARIANTS=0,207 timeout 300 ./kbench mla 128 8192 128 2>&1 | tee $OUT/mla_nt.log
cd $R
timeout 300 python3 tools/attn_decode_sweep.py 2>&1 | grep -v amdgpu | tee $OUT/attn_decode.log
timeout 300 python3 tools/qserve_bench.py 1 16 32 64 2>&1 | grep -v amdgpu | tee $OUT/qserve.log
