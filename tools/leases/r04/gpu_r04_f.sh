#!/bin/bash
# round 4, call F: QServe decode parity + timing, then the prefill-attention profile
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_f
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_qserve_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 300 python3 tools/qserve_bench.py 1 16 32 64 > $OUT/qserve.log 2>&1
cat $OUT/qserve.log
bash tools/gpu_profiles_r04.sh "prefill"
tail -70 $R/gpurun_out/r04/prof/attn_prefill.summary.txt
