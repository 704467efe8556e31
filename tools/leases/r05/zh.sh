#!/bin/bash
# r05 lease zh: fwd decode at short contexts, auto split count against explicit ones; attention parity
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zh
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_attention_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 900 python3 tools/row_sweep.py fwdsplit 2>&1 | grep "fwdsplit" | tee $OUT/fwdsplit.log
timeout 900 python3 tools/row_sweep.py fwdbs 2>&1 | grep "fwd decode" | tee $OUT/fwdbs.log
