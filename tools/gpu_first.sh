#!/bin/bash
# first on-box check: smoke, GPU parity tests, bench
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/sgl-kernel-xpu_amd/python
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/rocminfo.txt 2>&1
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -40 > gpurun_out/pytest.log; cat gpurun_out/pytest.log | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -3 gpurun_out/bench.log
