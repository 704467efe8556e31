#!/bin/bash
# r05 lease zn: 16-bit fused_experts at 61 .. 96 rows per expert on the four-wave 128-row streaming tile: MoE parity + token sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zn
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
MOE_FMTS=bf16 MOE_TS=128,192,256,288,320,352,384 timeout 600 python3 tools/row_sweep.py moe2 2>&1 | grep "fused_experts" | tee $OUT/moe2.log
