#!/bin/bash
# r05 lease zx: fp8_scaled_mm / int8_scaled_mm K-slice units: GEMM parity + the row sweep (rule, then forced 0 / 2 / 4 / 8 slices)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zx
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_gemm_gpu.py tests/test_cabi.py -m gpu -q -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
GEMM_MS=128,129,192,256,257,384,512,513,768,1024,1025 timeout 600 python3 tools/row_sweep.py gemm 2>&1 | grep "N=" | tee $OUT/sweep.log
for s in 0 2 4 8; do
  echo "== forced slices: $s"
  GEMM_MS=129,192,256,384,512,768,1024 GEMM_SPLITK=$s LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 600 python3 tools/row_sweep.py gemm 2>&1 | grep "N=" | tee $OUT/sweep_s$s.log
done
