#!/bin/bash
# r05 lease p: fwd prefill with an fp8 KV cache on the 128-row-block kernel: parity, timing next to the round-4 library
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_p
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1200 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for rep in 1 2; do
  echo "== r05"; ATTN_PREFILL_ONLY=fp8 timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
  echo "== r04"; ATTN_PREFILL_ONLY=fp8 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes_r04.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v "amdgpu\|ld.so"
done | tee $OUT/prefill_fp8.log
timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu | tee -a $OUT/prefill_fp8.log
