#!/bin/bash
# r05 lease b: self-resetting merge counters (no zeroing launch): parity, determinism, graph capture; then the in-kernel stamps,
# one probe per process (lease a: the combined run died with a memory fault)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_b
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_cabi.py tests/test_mla_prefill_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
MLA_GAUSS=100 timeout 120 ./kbench mla 128 8192 128
for p in 0 4 3 6 1 2 5 12 16; do
  MLA_GAUSS=100 MLA_STAMPS=$p timeout 100 ./kbench mla 128 8192 128 2>&1 | tail -3
done
} 2>&1 | tee $OUT/mla_stamps.log
