#!/bin/bash
# round 4, call I: MoE tile pipeline with bias + three-stage 128-row blocks: parity (whole MoE file), then timing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_i
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 600 python3 tools/moe_bench.py 256 512 768 1024 2048 > $OUT/moe.log 2>&1
cat $OUT/moe.log
