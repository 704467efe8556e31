#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/sgl-kernel-xpu_amd/python
timeout 900 python -m pytest tests/test_rope_quantv2_gpu.py -m gpu -q --timeout 600 2>&1 | tail -40 > gpurun_out/pytest_misc.log; tail -30 gpurun_out/pytest_misc.log
