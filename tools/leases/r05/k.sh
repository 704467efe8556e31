#!/bin/bash
# r05 lease k: fwd prefill at d = 256 and with softcap on the 128-row-block kernel: parity, timing next to the round-4 library;
# QServe prologue order A/B
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_k
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_attention_gpu.py tests/test_full_size_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for rep in 1 2; do
  echo "== r05"; timeout 300 python3 tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu
done | tee $OUT/prefill.log
LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so QSERVE_CFGS=43022,143022,23044,123044 timeout 600 python3 tools/qserve_bench.py 32 64 2>&1 | grep -v amdgpu | tee $OUT/qserve.log
