#!/bin/bash
# r05 lease zg: fwd prefill at head dims 96 / 192 on the 128-row-block kernel (inside the 128 / 256 LDS images): parity (the whole
# attention file), timing against the round-5 library before this change (general 16-row kernel) is the printed baseline of lease k
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zg
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
ATTN_PREFILL_ONLY=96 timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu | tee $OUT/prefill.log
