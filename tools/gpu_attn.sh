#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/sgl-kernel-xpu_amd/python
timeout 2400 python -m pytest tests/test_attention_gpu.py -m gpu -q --timeout 1200 -k "full_size or varlen or softcap or masked or golden or errors" 2>&1 | tail -60 > gpurun_out/pytest_attn.log; tail -50 gpurun_out/pytest_attn.log
