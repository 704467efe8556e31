#!/bin/bash
# r05 lease zb: K split of the MoE down projection (moe_persist KSPL = 2): parity, then fused_experts with / without it on one box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zb
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py tests/test_cabi.py -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for rep in 1 2; do
  for k in 1 0; do
    echo "== MOE_SPLITK=$k"
    MOE_SPLITK=$k MOE_BENCH_INT4_ONLY=1 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/moe_bench.py 512 768 1024 1536 2>&1 | grep "fused_experts T"
  done
done | tee $OUT/moe.log
