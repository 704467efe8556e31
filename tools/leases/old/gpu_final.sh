#!/bin/bash
# End-of-round validation on the GPU box: the whole -m gpu suite, smoke(), the default bench line, the timing scripts.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 3000 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python3 tools/moe_bench.py 1 16 64 256 2048 2>&1 | grep -v amdgpu.ids
timeout 300 python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
timeout 300 python3 tools/qserve_bench.py 2>&1 | grep -v amdgpu.ids
timeout 900 python3 bench.py 2>&1 | tail -1 > $OUT/bench.json
cut -c1-1200 $OUT/bench.json
