#!/bin/bash
# r05 lease zs: smoke + a quick slice of the -m gpu suite after a rebuild
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zs
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 120 python3 __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python3 -m pytest tests/test_cabi.py tests/test_graph_capture_gpu.py tests/test_determinism_gpu.py tests/test_sampling_gpu.py tests/test_qserve_gpu.py -m gpu -q 2>&1 | tail -3
