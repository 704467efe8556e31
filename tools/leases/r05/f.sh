#!/bin/bash
# r05 lease f: MLA (window fetch as one in-place statement): parity + stamps of the full loop; QServe W4A8 stream sweep at 64 rows
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_f
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_mla_decode_gpu.py tests/test_mla_prefill_gpu.py tests/test_determinism_gpu.py tests/test_graph_capture_gpu.py tests/test_qserve_gpu.py tests/test_norm_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
cd $R/sgl-kernel-xpu_amd/build
{
for rep in 1 2; do
  echo "== r05"; MLA_GAUSS=100 timeout 100 ./kbench mla 128 8192 128
  echo "== r04"; LD_PRELOAD=$PWD/libsglk_probes_r04.so MLA_GAUSS=100 timeout 100 ./kbench mla 128 8192 128
done
for p in 70 74 76; do
  MLA_GAUSS=100 MLA_STAMPS=$p timeout 100 ./kbench mla 128 8192 128 2>&1 | tail -5
done
} 2>&1 | tee $OUT/mla.log
cd $R
LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so QSERVE_CFGS=4320,4321,4340,4341,4241,4281,2341,2381,2441,2340,1441,1381 timeout 600 python3 tools/qserve_bench.py 32 64 2>&1 | grep -v amdgpu | tee $OUT/qserve.log
