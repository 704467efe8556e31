#!/bin/bash
# r05 lease zo: fwd with 17 - 64 packed rows on the decode kernel (16-row groups) and 65+ rows / explicit splits on the prefill
# kernel: parity of the attention file, the query-token sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zo
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 300 python3 tools/row_sweep.py fwd 2>&1 | grep "fwd bs" | tee $OUT/sweep.log; timeout 300 python3 tools/row_sweep.py prefill1 2>&1 | grep "fwd chunk" | tee -a $OUT/sweep.log
