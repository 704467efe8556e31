#!/bin/bash
# r05 lease c: the re-cut MLA tile loop (mla_rows128z_kernel): parity first, then A/B against the round-4 loop and stamps
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_c
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
MLA_GAUSS=100 MLA_AB=1 timeout 200 ./kbench mla 128 8192 128
for p in 90 70 74 73; do
  MLA_GAUSS=100 MLA_STAMPS=$p timeout 100 ./kbench mla 128 8192 128 2>&1 | tail -3
done
} 2>&1 | tee $OUT/mla_ab.log
