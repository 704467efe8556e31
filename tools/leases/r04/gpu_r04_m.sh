#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_m
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 600 python3 tools/moe_bench.py 256 384 512 768 > $OUT/moe.log 2>&1
head -4 $OUT/moe.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/moe_trace -- python3 $R/tools/moe_bench.py 512 > $OUT/moe_trace.log 2>&1
cd $R
python3 tools/summarize_prof.py $OUT/moe_trace | head -12 | cut -c1-200
find $OUT/moe_trace -name "*_kernel_trace.csv" -delete
