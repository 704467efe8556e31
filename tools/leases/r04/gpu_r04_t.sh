#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_t
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 2000 python3 -m pytest tests/test_attention_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for w in 4 8 4 8; do
  echo "== waves $w"
  SPLITS=0,1,2,4 ATTN_WAVES=$w LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/attn_decode_sweep.py 64:bf16 128:fp8 128:bf16 2>&1 | grep -v amdgpu
done > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
