#!/bin/bash
# r05 lease e: A/B of the MLA decode against the round-4 library on one box (LD_PRELOAD), per-wave stamps, parity
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_e
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
for rep in 1 2 3; do
  echo "== r05"; MLA_GAUSS=100 timeout 100 ./kbench mla 128 8192 128
  echo "== r04"; LD_PRELOAD=$PWD/libsglk_probes_r04.so MLA_GAUSS=100 timeout 100 ./kbench mla 128 8192 128
done
for p in 70 74 73 75; do
  MLA_GAUSS=100 MLA_STAMPS=$p timeout 100 ./kbench mla 128 8192 128 2>&1 | tail -5
done
echo "== other shapes r05 / r04"
for cfg in "64 1024 128" "64 4096 128" "32 8192 128" "256 2048 128" "64 4096 96"; do
  MLA_GAUSS=100 timeout 100 ./kbench mla $cfg | tail -1
  LD_PRELOAD=$PWD/libsglk_probes_r04.so MLA_GAUSS=100 timeout 100 ./kbench mla $cfg | tail -1
done
} 2>&1 | tee $OUT/mla_ab.log
