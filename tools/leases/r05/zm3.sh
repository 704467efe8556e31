#!/bin/bash
# r05 lease zm3: apply_shuffle_mul_sum over (token, column chunk) workgroups: MoE parity + decode-sized fused_experts at two shapes
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zm3
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py tests/test_determinism_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
MOE_TS=1,2,4,8,16,64,512,2048 MOE_SHAPE=256,7168,2048,8 timeout 600 python3 tools/row_sweep.py moe 2>&1 | grep "fused_experts" | tee $OUT/moe_dsv3.log
MOE_TS=1,2,4,8,16,64,512,2048 timeout 600 python3 tools/row_sweep.py moe 2>&1 | grep "fused_experts" | tee $OUT/moe_mixtral.log
