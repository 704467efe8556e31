#!/bin/bash
# r05 lease zj: the gpt-oss swiglu in the grouped GEMMs' epilogues (16-bit: the reference's activation_type 2 with fuse_act; 4-bit:
# authored activation 5): parity of the MoE file + host-visible sequence, gpt-oss fused_experts timing against the build before
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zj
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py tests/test_graph_capture_gpu.py tests/test_cabi.py -m gpu -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
