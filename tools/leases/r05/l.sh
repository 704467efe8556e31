#!/bin/bash
# r05 lease l: fwd prefill with softcap on the 128-row-block kernel: parity; d = 128 / 64 / chunk timing next to the round-4
# library on one box; QServe defaults
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_l
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py tests/test_qserve_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for rep in 1 2; do
  echo "== r05"; timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
  echo "== r04"; LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes_r04.so timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v "amdgpu\|ld.so"
done | tee $OUT/prefill.log
timeout 600 python3 tools/qserve_bench.py 1 16 32 64 2>&1 | grep -v amdgpu | tee $OUT/qserve.log
