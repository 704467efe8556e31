#!/bin/bash
# r05 lease zi: fwd chunks of long sequences: the auto split count against explicit ones; attention parity
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zi
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_attention_gpu.py -m gpu -q -k "split or chunk or prefill" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 900 python3 tools/row_sweep.py prefillsplit 2>&1 | grep "prefillsplit" | cut -c1-100 | tee $OUT/prefillsplit.log
timeout 600 python3 tools/row_sweep.py fwd 2>&1 | grep "fwd bs16" | tee $OUT/fwd.log
