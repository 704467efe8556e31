#!/bin/bash
# round 4, call D: flash_mla_decode with the split merge inside the kernel - parity, determinism, graph capture, timing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_d
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd $R
timeout 1200 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_cabi.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
{
  MLA_GAUSS=100 timeout 200 $K mla 128 8192 128 -1 1 2 4
  MLA_GAUSS=100 timeout 200 $K mla 32 8192 128 -1
  MLA_GAUSS=100 timeout 200 $K mla 128 8192 64 -1
} > $OUT/kbench.log 2>&1
cat $OUT/kbench.log
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
