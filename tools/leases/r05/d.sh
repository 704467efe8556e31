#!/bin/bash
# r05 lease d: MLA epilogue in fragment order + LDS-staged rows: parity, then A/B against the round-4 library (LD_PRELOAD), stamps
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_d
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_cabi.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
for rep in 1 2 3; do
  echo "== r05"; MLA_GAUSS=100 timeout 100 ./kbench mla 128 8192 128
  echo "== r04"; LD_PRELOAD=$PWD/libsglk_probes_r04.so MLA_GAUSS=100 timeout 100 ./kbench mla 128 8192 128
done
for p in 70 74; do
  MLA_GAUSS=100 MLA_STAMPS=$p timeout 100 ./kbench mla 128 8192 128 2>&1 | tail -3
done
echo "== other shapes r05 / r04"
for hh in 128 96; do for ss in 1024 4096; do
  MLA_GAUSS=100 timeout 100 ./kbench mla 64 $ss $hh | tail -1
  LD_PRELOAD=$PWD/libsglk_probes_r04.so MLA_GAUSS=100 timeout 100 ./kbench mla 64 $ss $hh | tail -1
done; done
} 2>&1 | tee $OUT/mla_ab.log
