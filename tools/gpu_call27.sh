#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 3000 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2048 2>&1 | grep -v amdgpu.ids
timeout 900 python3 bench.py 2>&1 | tail -1 > gpurun_out/r02_bench_c27.json; cut -c1-1500 gpurun_out/r02_bench_c27.json
