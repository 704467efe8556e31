#!/bin/bash
# r05 lease q: MoE tile pipeline takes the 1..64-row remainders itself when all tiles fit one round (own_rem): parity, timing next
# to the round-4 library on one box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_q
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_moe_gpu.py tests/test_full_size_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for rep in 1 2; do
  echo "== r05"; MOE_BENCH_INT4_ONLY=1 timeout 300 python3 tools/moe_bench.py 128 256 384 512 768 1024 2>&1 | grep "fused_experts T"
  echo "== r04"; MOE_BENCH_INT4_ONLY=1 LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes_r04.so timeout 300 python3 tools/moe_bench.py 128 256 384 512 768 1024 2>&1 | grep "fused_experts T"
done | tee $OUT/moe.log
