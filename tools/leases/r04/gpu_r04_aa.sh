#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_aa
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for rep in 1 2; do
  for v in old A cur; do
    echo "== $v (rep $rep)"
    if [ $v = cur ]; then
      timeout 600 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
    else
      LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_$v.so timeout 600 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
    fi
  done
done | tee $OUT/prefill_ab.log
timeout 1500 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -n 4 2>&1 | tail -4
