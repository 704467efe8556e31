#!/bin/bash
# round 4, call G: whole GPU suite + full bench
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_g
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
