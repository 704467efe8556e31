#!/bin/bash
# r05 lease x: MLA parity with the 16-wide QK^T selected by launch size (wide-table decode cases, prefill head-slot cases)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_x
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_full_size_gpu.py tests/test_cabi.py -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
