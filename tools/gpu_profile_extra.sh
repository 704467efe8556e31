#!/bin/bash
# rocprofv3 kernel-trace stats of the `extra` part of bench.py (MLA decode, fwd, fused_experts, HBM streams)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_extra
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cat {} \;
