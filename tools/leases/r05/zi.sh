#!/bin/bash
# r05 lease zi: int4 expansion of the MoE tile pipeline with packed multiplies (v_pk_mul_f32: 19 instead of 23 vector instructions per
# dword of codes) against the build before (LD_PRELOAD of build/libsglk_prev.so), interleaved on one box; parity of the MoE file
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zi
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for rep in 1 2 3; do
  echo "== packed"; MOE_BENCH_INT4_ONLY=1 timeout 300 python3 tools/moe_bench.py 512 2048 4096 2>&1 | grep "fused_experts T"
  echo "== before"; MOE_BENCH_INT4_ONLY=1 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_prev.so timeout 300 python3 tools/moe_bench.py 512 2048 4096 2>&1 | grep "fused_experts T"
done | tee $OUT/moe.log
