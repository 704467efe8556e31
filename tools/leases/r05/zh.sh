#!/bin/bash
# r05 lease zh: fwd decode at head dims 96 / 192 on the independent-wave kernel (inside its d = 128 / 256 forms): parity, timing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zh
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_attention_gpu.py -m gpu -q -k "decode or other_head_dims or golden" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 300 python3 tools/attn_decode_sweep.py 96:bf16 192:bf16 128:bf16 2>&1 | grep -v amdgpu | tee $OUT/sweep.log
