#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/sgl-kernel-xpu_amd/python
timeout 1800 python -m pytest tests/test_moe_gpu.py -m gpu -q --timeout 900 2>&1 | tail -60 > gpurun_out/pytest_moe.log; tail -50 gpurun_out/pytest_moe.log
