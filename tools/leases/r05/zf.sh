#!/bin/bash
# r05 lease zf: fwd decode over 16-token pages on the decode kernel (two page ids per tile): attention parity + the layout / page sweep
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zf
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_attention_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 900 python3 tools/row_sweep.py fwdcfg 2>&1 | grep "fwdcfg" | tee $OUT/fwdcfg.log
