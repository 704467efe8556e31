#!/bin/bash
# r05 lease zy: MoE parity + fused_experts sweeps over token counts for 16-bit and mxfp4 weights (ZY_INT4=1: the int4 path too)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zy
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py tests/test_determinism_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 900 python3 tools/row_sweep.py moe2 2>&1 | grep "fused_experts" | tee $OUT/moe2.log
if [ -n "$ZY_INT4" ]; then timeout 900 python3 tools/row_sweep.py moe 2>&1 | grep "fused_experts" | tee $OUT/moe.log; fi
