#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/sgl-kernel-xpu_amd/python
timeout 1500 python -m pytest tests/test_mla_decode_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -40 > gpurun_out/pytest_mla.log; tail -30 gpurun_out/pytest_mla.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_mla.log 2>&1; tail -3 gpurun_out/bench_mla.log
