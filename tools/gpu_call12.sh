#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c12
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 600 python3 tools/w4a16_bench.py > $OUT/w4a16_bench.log 2>&1
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2048 > $OUT/moe_bench.log 2>&1
cat $OUT/w4a16_bench.log $OUT/moe_bench.log
timeout 1500 python3 -m pytest tests/test_moe_gpu.py -x -q -m gpu -k "w4a16 or golden or fused_experts or mxfp4" > $OUT/pytest_moe.log 2>&1
tail -5 $OUT/pytest_moe.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/moe_trace -- python3 $R/tools/moe_bench.py 64 > $OUT/moe_trace.log 2>&1
cd $R; python3 tools/summarize_prof.py $OUT/moe_trace | cut -c1-200 | head -20
