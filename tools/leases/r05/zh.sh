#!/bin/bash
# r05 lease zh: flash_mla_decode auto split count against explicit ones; MLA parity
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zh
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -k "mla" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 900 python3 tools/row_sweep.py mlasplit 2>&1 | grep "mlasplit" | cut -c1-90 | tee $OUT/mlasplit.log
