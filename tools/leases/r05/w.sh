#!/bin/bash
# r05 lease w: kQ16 selected by launch size: MLA parity; flash_mla_prefill on the 16-wide / 32-wide QK^T
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_w
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
MLA_VARIANTS=216,232 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 600 python3 tools/mla_prefill_bench.py 2>&1 | grep -v amdgpu | tee $OUT/prefill.log
