#!/bin/bash
# r05 lease zv: fp8_blockwise_scaled_mm K-slice units (few rows over a deep K): GEMM parity + the row sweep across the dispatch
# boundaries; ZV_FORCED=1 adds the sweep with 0 / 2 / 4 / 8 forced slices (diagnostic build)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zv
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_gemm_gpu.py -m gpu -q -x -k "blockwise" > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
GEMM_MS=40,48,56,64,65,96,128,129,192,256,257,384,512,513,768,1024,1025 timeout 600 python3 tools/row_sweep.py gemmbw 2>&1 | grep "N=" | tee $OUT/sweep.log
if [ -n "$ZV_FORCED" ]; then
  for s in 0; do
    echo "== forced slices: $s"
    GEMM_MS=40,48,56,64,65,128,192,256,512 GEMM_SPLITK=$s LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 600 python3 tools/row_sweep.py gemmbw 2>&1 | grep "N=" | tee $OUT/sweep_s$s.log
  done
fi
