#!/bin/bash
# r05 lease zu: sampling ops recorded into a HIP graph (device-resident generator state) + sweeps of the small ops
# (sampling at vocab 128256; rope / qk-norm / cache store / merge / routers / align / per-tensor quant / awq over 1..16384 tokens)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zu
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_sampling_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 300 python3 tools/row_sweep.py sample 2>&1 | tail -8 | tee $OUT/sample.log
if [ -n "$ZU_ELEM3" ]; then timeout 600 python3 tools/row_sweep.py elem3 2>&1 | tail -70 | tee $OUT/elem3.log; fi
