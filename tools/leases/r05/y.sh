#!/bin/bash
# r05 lease y: restored tree after the container change: full -m gpu suite, smoke, bench line
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_y
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1700 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 120 python3 __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
