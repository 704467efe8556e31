#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_k
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log; grep -n "AssertionError: (" $OUT/pytest.log | head -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/moe_trace -- python3 $R/tools/moe_bench.py 512 > $OUT/moe_trace.log 2>&1
cd $R
python3 tools/summarize_prof.py $OUT/moe_trace | head -30
find $OUT/moe_trace -name "*_kernel_trace.csv" -delete
LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/moe_gemm_split.py 512 1024 > $OUT/split.log 2>&1
cat $OUT/split.log | tail -20
