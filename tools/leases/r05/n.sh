#!/bin/bash
# r05 lease n: fwd prefill at d = 256 on the 128-row-block kernel (swizzle keys generalised): parity, timing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_n
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for rep in 1 2; do
  timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done | tee $OUT/prefill.log
